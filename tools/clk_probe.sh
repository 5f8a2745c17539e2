python - <<'PY' &
import sys, time, torch
sys.path.insert(0,'.')
import bench
dev=torch.device('cuda:0')
wl=bench.Workload('cfg2',dev,seed=1)
t0=time.time()
with torch.no_grad():
    while time.time()-t0<6:
        for _ in range(50): wl.model(wl.xd, wl.params)
        torch.cuda.synchronize()
PY
sleep 3
for i in 1 2 3; do rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|fclk\|mclk" | head -4; sleep 0.7; done
wait
echo idle:; rocm-smi --showclocks 2>/dev/null | grep -i "sclk" | head -2
