"""Which ATen ops (not this library's kernels) run in one eager training step, with counts and input shapes:
the launch-diet worklist.  usage: python profiles/aten_ops_step.py [size]"""
import sys, os, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import coma_unet_amd as cu
from coma_unet_amd.synthetic import make_batch
from coma_unet_amd.train import train_step, make_optimizer
S = (int(sys.argv[1]) if len(sys.argv) > 1 else 64,) * 3
dev = torch.device("cuda")
torch.manual_seed(0)
m = cu.build_model(volume_shape=S, compute_dtype=torch.bfloat16, static_prompts=True).to(dev); m.set_save_attn(None); m.train(True)
crit = cu.build_reference_criterion(dev); opt = make_optimizer(m, 1e-3)
b = make_batch(2, S, seed=1); batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()}
batch["roi_pred_dicts"] = m._priors(b["roi_pred_dicts"], 2, dev)
for _ in range(3): train_step(m, crit, opt, batch)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    train_step(m, crit, opt, batch)
    torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith("aten::") and \
            (e.cpu_parent is None or not e.cpu_parent.name.startswith("aten::")):
        # top-level aten op under a python/autograd frame
        stack = [s for s in (e.stack or []) if "coma_unet_amd" in s]
        cnt[(e.name, str(e.input_shapes)[:70], stack[0].split("coma_unet_amd/")[-1][:60] if stack else (e.cpu_parent.name[:60] if e.cpu_parent else "-"))] += 1
for (name, shp, where), n in sorted(cnt.items(), key=lambda kv: -kv[1])[:90]:
    print(f"{n:4d} {name:28s} {where:62s} {shp}")
ks = collections.Counter()
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CUDA:
        ks[e.name.replace('at::native::', '').replace('void ', '')[:110]] += 1
print("--- device kernels:", sum(ks.values()))
for k, n in ks.most_common(70):
    print(f"{n:4d} {k}")
