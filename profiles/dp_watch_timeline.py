"""When do the gradient buckets of the graph-replayed step become sendable?  One process, one GPU, no collective: the step
is captured the way `bench.py --gpus N` (default `--dp torch`) captures it -- forward + backward in the graph with
data_parallel.GraphBucketWatch recording an external event behind every completed bucket -- and after each replay an
auxiliary stream waits for every event and stamps the time.  Output: the probe's verdict for this machine, the buckets
(MB, when ready, how much of the replayed graph was still ahead) and how many MB are ready at which point of the step.

    COMA_DP_GRAPH_OVERLAP=force python profiles/dp_watch_timeline.py [size]
"""
import os
import sys
import time

import torch

os.environ.setdefault("COMA_DP_GRAPH_OVERLAP", "force")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import coma_unet_amd as cu  # noqa: E402
from coma_unet_amd.data_parallel import GradReducer, GraphBucketWatch  # noqa: E402
from coma_unet_amd.synthetic import make_batch  # noqa: E402
from coma_unet_amd.train import GraphedTrainStep, make_optimizer  # noqa: E402

dev = torch.device("cuda")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
S = (n,) * 3
print("probe:", GraphBucketWatch.probe(dev))
torch.manual_seed(0)
m = cu.build_model(volume_shape=S, compute_dtype=torch.bfloat16, static_prompts=True).to(dev)
m.set_save_attn(None)
m.train(True)
crit = cu.build_reference_criterion(dev)
opt = make_optimizer(m, 1e-3)
b = make_batch(2, S, seed=1)
batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()}
batch["roi_pred_dicts"] = m._priors(b["roi_pred_dicts"], 2, dev)
red = GradReducer(opt, overlap=False)
names = {id(p): k for k, p in m.named_parameters()}
marks = []
_mark = GraphBucketWatch.mark


def mark(self, p):
    marks.append(id(p))
    _mark(self, p)


GraphBucketWatch.mark = mark
step = GraphedTrainStep(m, crit, opt, batch, warmup=3, reducer=red)
w = step.watch
print("watch:", None if w is None else f"{len(w.bounds)} buckets, {len(w.groups)} event groups, probe {red.watch_probe}")
if w is None:
    sys.exit(0)
if "--marks" in sys.argv:
    order = {pid: i for i, pid in enumerate(marks)}
    print(f"{len(marks)} write-through gradients announced during the captured backward")
    for bi, (s0, e0) in enumerate(w.bounds):
        ps = [p for p in opt._flat_params if w._p2b[id(p)] == bi]
        seen = sorted((order[id(p)], names.get(id(p), "?")) for p in ps if id(p) in order)
        never = [names.get(id(p), "?") for p in ps if id(p) not in order]
        first = f"first announced #{seen[0][0]} {seen[0][1]}; last three: {seen[-3:]}" if seen else "none announced"
        print(f"bucket {bi}: {(e0 - s0) * 4 / 2 ** 20:.1f} MB, {len(ps)} parameters; {first}; never announced: {never[:6]}{' ...' if len(never) > 6 else ''} ({len(never)})")
aux = torch.cuda.Stream()
for it in range(4):
    t0 = torch.cuda.Event(enable_timing=True)
    t1 = torch.cuda.Event(enable_timing=True)
    stamps = []
    torch.cuda.synchronize()
    t0.record()
    h0 = time.perf_counter()
    step.graph.replay()
    h1 = time.perf_counter()
    t1.record()
    for todo, ev in w.groups:
        w.wait(ev, aux)
        e = torch.cuda.Event(enable_timing=True)
        e.record(aux)
        stamps.append(e)
    red.reduce_flat_and_step()          # (one rank: the fused AdamW step)
    torch.cuda.synchronize()
    if it < 2:
        continue
    total = t0.elapsed_time(t1)
    print(f"replay {it}: graph (forward + backward) {total:.2f} ms on the device, {1e3 * (h1 - h0):.2f} ms inside graph.replay() on the host")
    mb_all = sum(e0 - s0 for s0, e0 in w.bounds) * 4 / 2 ** 20
    cum = 0.0
    for (todo, _), e in zip(w.groups, stamps):
        mb = sum(w.bounds[i][1] - w.bounds[i][0] for i in todo) * 4 / 2 ** 20
        cum += mb
        t = t0.elapsed_time(e)
        print(f"  buckets {todo}: {mb:7.1f} MB ready at {t:6.2f} ms ({total - t:5.2f} ms of the graph ahead)   cumulative {cum:6.1f} / {mb_all:.1f} MB")
    late = [i for i in range(len(w.bounds)) if all(i not in todo for todo, _ in w.groups)]
    print(f"  behind the graph: buckets {late}: {sum(w.bounds[i][1] - w.bounds[i][0] for i in late) * 4 / 2 ** 20:.1f} MB")
