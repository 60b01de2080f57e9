"""Per-workgroup wall-clock stamps of the 256x256 GEMM (prologue done, k-loop done, epilogue issued, stores\ndrained): where a tile spends its time.  usage: python tools/gemm256_stamps.py"""
import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import gemm, _lib
dev = torch.device("cuda:0")
lib = _lib.kernels()
lib.dtf_gemm256_set_stamps.argtypes = [ctypes.c_void_p]
for (M, N, K) in [(32768, 3072, 768), (2048, 1024, 768), (512, 512, 768)]:
    a = torch.randn(M, K, device=dev).bfloat16(); b = torch.randn(N, K, device=dev).bfloat16()
    for _ in range(3): gemm.gemm256_nt(a, b)
    nblk = (M // 256) * (N // 256)
    st = torch.zeros(nblk * 4, dtype=torch.int64, device=dev)
    lib.dtf_gemm256_set_stamps(st.data_ptr())
    gemm.gemm256_nt(a, b); torch.cuda.synchronize()
    lib.dtf_gemm256_set_stamps(None)
    s = st.view(nblk, 4).cpu().double()
    t0 = s[:, 0].min()
    d = (s - t0) / 100.0   # us
    print(M, N, K, "blocks", nblk)
    for name, col in (("start", 0), ("prologue_done", 1), ("kloop_done", 2), ("epilogue_done", 3)):
        print(f"  {name:14s} min {float(d[:,col].min()):8.2f} median {float(d[:,col].median()):8.2f} max {float(d[:,col].max()):8.2f}")
    print("  epilogue issue time (us) median %.2f ; start->end median %.2f" % (float(s[:, 2].median()) / 100.0, float((s[:, 3] - s[:, 0]).median()) / 100.0))
    continue
    first = s[:min(256, nblk)]
    seg1 = first[:, 1:] - first[:, :-1]
    print("  first-wave blocks median: prologue %.2f  kloop %.2f  epilogue %.2f" % tuple(float(x) / 100.0 for x in seg1.median(0).values))
