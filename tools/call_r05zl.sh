# r05zl: cLN backward after the register diet (products of the first phase formed again instead of kept): 16 waves x 4 rounds against 8 x 8 at C = 512
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
( timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "cln" 2>&1 | tail -2 )
for nw in 16 8 16 8; do echo "== SEPK_CLN_BWD_NW=$nw"; SEPK_CLN_BWD_NW=$nw timeout 300 python /dev/stdin <<'P'
import torch, sepkernels
K = sepkernels.HipBackend(); dev = "cuda"
for (B, C, T, ldt) in [(16, 512, 3999, 4096), (16, 128, 3999, 4096), (16, 256, 3999, 4096)]:
    x = torch.randn(B, C, ldt, device=dev); dy = torch.randn(B, C, ldt, device=dev)
    g = torch.randn(C, device=dev); b = torch.randn(C, device=dev); al = torch.tensor([0.25], device=dev)
    y = torch.empty_like(x); dx = torch.empty_like(x); mean = torch.empty(B, ldt, device=dev); rstd = torch.empty(B, ldt, device=dev)
    ws = torch.empty((K.cln_ws_bytes(B, C, T, ldt) + 7) // 8, device=dev, dtype=torch.float64)
    pg, pb, pa = (torch.empty(B, C, device=dev) for _ in range(3))
    def run(fn, n=20):
        for _ in range(3): fn()
        torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
    tf = run(lambda: K.cln_fwd(x, g, b, y, mean, rstd, ws, B, C, T, ldt, 1e-12, alpha=al))
    tb = run(lambda: K.cln_bwd(dy, x, g, mean, rstd, dx, pg, pb, ws, B, C, T, ldt, 1e-12, alpha=al, dalpha_part=pa))
    print("B%d C%d T%d  fwd %.1f us  bwd %.1f us" % (B, C, T, tf, tb))
P
done
