import os, sys, time, torch, torch.distributed as dist
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import coma_unet_amd as cu
from coma_unet_amd.synthetic import make_batch
from coma_unet_amd.train import train_step, make_optimizer, GraphedTrainStep
from coma_unet_amd.data_parallel import GradReducer, broadcast_module
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29577", rank=0, world_size=1, device_id=torch.device("cuda",0))
S=(64,)*3; dev=torch.device("cuda")
b = make_batch(2, S, seed=1); batch = {k:(v.to(dev) if torch.is_tensor(v) else v) for k,v in b.items()}
for mode in ("eager", "graph"):
    torch.manual_seed(0)
    m = cu.build_model(volume_shape=S, compute_dtype=torch.bfloat16, static_prompts=True).to(dev); m.set_save_attn(None); m.train(True)
    batch["roi_pred_dicts"] = m._priors(b["roi_pred_dicts"], 2, dev)
    crit = cu.build_reference_criterion(dev); opt = make_optimizer(m, 1e-3)
    red = GradReducer(opt); red.world = 2
    if mode == "eager":
        for _ in range(3): train_step(m, crit, opt, batch, red)
        f = lambda: train_step(m, crit, opt, batch, red)
    else:
        g = GraphedTrainStep(m, crit, opt, batch, warmup=3, reducer=red); f = lambda: g()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(10): f()
    torch.cuda.synchronize(); print(mode, (time.perf_counter()-t)/10*1e3, "ms/step")
dist.destroy_process_group()
