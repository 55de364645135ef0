import sys, os, cProfile, pstats, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import coma_unet_amd as cu
from coma_unet_amd.synthetic import make_batch
from coma_unet_amd.train import train_step, make_optimizer
S=(128,)*3
dev=torch.device("cuda")
torch.manual_seed(0)
m = cu.build_model(volume_shape=S, compute_dtype=torch.bfloat16, static_prompts=True).to(dev); m.set_save_attn(None); m.train(True)
crit = cu.build_reference_criterion(dev); opt = make_optimizer(m, 1e-3)
b = make_batch(2, S, seed=1); batch = {k:(v.to(dev) if torch.is_tensor(v) else v) for k,v in b.items()}
batch["roi_pred_dicts"] = m._priors(b["roi_pred_dicts"], 2, dev)
for _ in range(3): train_step(m, crit, opt, batch)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(5): train_step(m, crit, opt, batch)
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
