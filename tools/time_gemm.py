"""Times d3p_gemm_f32 (fp32 MFMA) on the GEMM shapes of the VAE step (BASELINE config 5: 784 -> 400 -> 50, B = 4096):
HIP events around 20 launches each; prints microseconds and TFLOP/s (fp32 MFMA peak of MI355X: 157)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import d3p_amd._lib as lib

B, D, H, Z = 4096, 784, 400, 50
L = lib.load()
lib.require_device()
g = torch.Generator().manual_seed(0)
X = torch.randn(B, D, generator=g).cuda()
W1 = torch.randn(D, H, generator=g).cuda()
h = torch.randn(B, H, generator=g).cuda()
V2 = torch.randn(H, D, generator=g).cuda()
da = torch.randn(B, D, generator=g).cuda()
Wl = torch.randn(H, Z, generator=g).cuda()
dz = torch.randn(B, Z, generator=g).cuda()
V1 = torch.randn(Z, H, generator=g).cuda()

# (name, A, a_sm, a_sk, B, b_sk, b_sn, M, N, K)
cases = [
    ("h1  = X W1        (4096 x 400 x 784, NN)", X, D, 1, W1, H, 1, B, H, D),
    ("a   = h2 V2       (4096 x 784 x 400, NN)", h, H, 1, V2, D, 1, B, D, H),
    ("dh2 = da V2^T     (4096 x 400 x 784, NT)", da, D, 1, V2, 1, D, B, H, D),
    ("dV2 = h2^T da     (400 x 784 x 4096, TN)", h, 1, H, da, D, 1, H, D, B),
    ("dW1 = X^T dh1     (784 x 400 x 4096, TN)", X, 1, D, h, H, 1, D, H, B),
    ("zl  = h1 Wl       (4096 x 50 x 400, NN)", h, H, 1, Wl, Z, 1, B, Z, H),
    ("h2  = z V1        (4096 x 400 x 50, NN)", dz, Z, 1, V1, H, 1, B, H, Z),
    ("dz  = dh2 V1^T    (4096 x 50 x 400, NT)", h, H, 1, V1, 1, H, B, Z, H),
    ("dWl = h1^T dz     (400 x 50 x 4096, TN)", h, 1, H, dz, Z, 1, H, Z, B),
]
for name, A, a_sm, a_sk, Bm, b_sk, b_sn, M, N, K in cases:
    out = torch.empty(M, N, device="cuda")
    def run():
        lib.check(L.d3p_gemm_f32(lib.stream_ptr(), lib.ptr(A), a_sm, a_sk, lib.ptr(Bm), b_sk, b_sn, lib.ptr(out), N, M, N, K, None, 1.0, 0))
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    ref = (A.double() if a_sk == 1 else A.double().t()) @ (Bm.double() if b_sn == 1 else Bm.double().t())
    err = float((out.double() - ref).abs().max()) / float(ref.abs().max())
    print("%-44s %7.1f us  %6.1f TFLOP/s  rel err %.1e" % (name, us, 2.0 * M * N * K / us / 1e6, err))
