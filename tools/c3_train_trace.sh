export TMPDIR=/tmp
cd /tmp
cat > /tmp/c3.py <<'PY'
import sys, time, torch
sys.path.insert(0, "/root/repo")
from egnn_pytorch_amd import EGNN_Network
torch.manual_seed(0)
net = EGNN_Network(depth=3, dim=128, num_nearest_neighbors=32).cuda()
feats = torch.randn(64, 1024, 128, device="cuda", requires_grad=True)
coors = torch.randn(64, 1024, 3, device="cuda", requires_grad=True)
mask = torch.ones(64, 1024, dtype=torch.bool, device="cuda")
for it in range(4):
    f, c = net(feats, coors, mask=mask)
    (f.square().mean() + c.square().mean()).backward()
    net.zero_grad(); feats.grad = None; coors.grad = None
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/c3trace -o t --output-format csv -- python /tmp/c3.py > /root/repo/gpurun_out/c3trace.log 2>&1
cd /root/repo
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/c3trace/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total {tot/4e6:.2f} ms per step")
for r in rows[:45]:
    print(f'{float(r["TotalDurationNs"])/4e6:8.3f} ms/step {int(r["Calls"])/4:6.1f} calls/step {float(r["AverageNs"])/1e3:8.1f} us  {r["Name"][:110]}')
PY
rm -f gpurun_out/c3trace/*kernel_trace.csv gpurun_out/c3trace/*/*kernel_trace.csv
