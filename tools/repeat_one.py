"""Run ONE pytest test (node id) N times inside one process: `python tools/repeat_one.py <node id> [N=20]`.
(For races that show on some boxes only: the GPU box allows one test process at a time, so the repetition happens in-process.)"""
import sys


def main(argv):
    import pytest
    node, n = argv[1], int(argv[2]) if len(argv) > 2 else 20
    for i in range(n):
        rc = pytest.main(["-x", "-q", "-p", "no:cacheprovider", node])
        if rc != 0:
            print(f"repetition {i + 1} of {n} failed (exit code {int(rc)})")
            return int(rc)
    print(f"{n} repetitions of {node}: all passed")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
