# r05zv: HBM traffic of the chained cLN (FETCH_SIZE x 2 / WRITE_SIZE, as tools/pmc_traffic.py corrects them) at B = 16, C = 512, T = 3999
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R/dnn-based_source_separation_amd/src
cat > /tmp/cln_once.py <<'P'
import torch, sepkernels
K = sepkernels.HipBackend(); dev = "cuda"
B, C, T, ldt = 16, 512, 3999, 4096
x = torch.randn(B, C, ldt, device=dev); dy = torch.randn(B, C, ldt, device=dev)
g = torch.randn(C, device=dev); b = torch.randn(C, device=dev); al = torch.tensor([0.25], device=dev)
y = torch.empty_like(x); dx = torch.empty_like(x); mean = torch.empty(B, ldt, device=dev); rstd = torch.empty(B, ldt, device=dev)
ws = torch.empty((K.cln_ws_bytes(B, C, T, ldt) + 7) // 8, device=dev, dtype=torch.float64)
pg, pb, pa = (torch.empty(B, C, device=dev) for _ in range(3))
for _ in range(3):
    K.cln_fwd(x, g, b, y, mean, rstd, ws, B, C, T, ldt, 1e-12, alpha=al)
    K.cln_bwd(dy, x, g, mean, rstd, dx, pg, pb, ws, B, C, T, ldt, 1e-12, alpha=al, dalpha_part=pa)
torch.cuda.synchronize()
P
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  for ch in 1 0; do
    SEPK_CLN_CHAIN=$ch timeout 200 rocprofv3 --pmc $c -d /tmp/pmc_${c}_$ch -- python /tmp/cln_once.py > /tmp/pmc.log 2>&1; tail -2 /tmp/pmc.log | cut -c1-120
    echo "== $c SEPK_CLN_CHAIN=$ch"; python $R/tools/pmc_summary.py /tmp/pmc_${c}_$ch "%cln%" | cut -c1-110
  done
done
