#!/bin/bash
# Same-box A/B of two BUILDS of the library (run through gpurun): the working tree against another commit, each with its own
# libd3p_hip.so.  Round 3's lesson: a run-time switch inside a kernel is not an A/B -- the compiler emits
# different code for BOTH settings as soon as the switch exists (0.14 us per step for one `if (a.dbg & ...)` around two loads),
# and boxes differ less than that (0.5 %), so "it is the box" is not an explanation either.
# usage (in the container):  bash tools/ab_builds.sh prepare <commit>      # exports <commit> to ab_other/ and builds it there
#        (on the GPU box):   gpurun -- 'bash tools/ab_builds.sh run [time_chained.py arguments]'
#        (afterwards):       bash tools/ab_builds.sh clean
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
case "$1" in
prepare)
    rm -rf "$R/ab_other"; mkdir -p "$R/ab_other"
    git -C "$R" archive "$2" | tar -x -C "$R/ab_other"
    (cd "$R/ab_other" && python3 -c "import d3p_amd._lib as L; L.build()")
    ;;
run)
    shift
    for i in 1 2 3; do
        echo "== other (ab_other/)"; (cd "$R/ab_other" && python3 tools/time_chained.py "$@" | tail -1)
        echo "== working tree";      (cd "$R" && python3 tools/time_chained.py "$@" | tail -1)
    done
    ;;
clean) rm -rf "$R/ab_other" ;;
*) echo "usage: $0 prepare <commit> | run [args] | clean"; exit 2 ;;
esac
