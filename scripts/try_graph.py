"""Development aid: the train step replayed from a captured HIP graph vs launched eagerly (are the inter-kernel gaps worth a graph?)."""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
args = types.SimpleNamespace(workload="train", batch=0, steps=20, warmup=5, precision="bf16x3", no_cpu_baseline=True, gpus=1)
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
tr, heads, imgs, labels, w, acc = bench.make_train_state(args, 1, 0, 256, dev)
step = lambda: tr.train_step(heads, imgs, labels, w, acc)
for _ in range(5): step()
torch.cuda.synchronize()
def timeit(fn, n=30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize()
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print("eager : %.3f ms/step" % timeit(step))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
try:
    with torch.cuda.graph(g):
        step()
    torch.cuda.synchronize()
    print("graph : %.3f ms/step" % timeit(g.replay))
    print("eager : %.3f ms/step" % timeit(step))
except Exception as e:
    print("capture failed:", repr(e)[:300])
