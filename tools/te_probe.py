"""Debug probe for the graph-captured engine step (usage: te_probe.py <mode>)."""
import copy, sys, os, torch
sys.path.insert(0, os.getcwd())
from tests.test_train_engine import CONFIG
from unidefense_amd.engine import get_engine
from unidefense_amd.model import perturb
mode = sys.argv[1]
cfg = copy.deepcopy(CONFIG)
cfg["model"]["drop_rate"] = 0.0
cfg["model"]["drop_connect_rate"] = 0.0
if "force_down" in mode:
    perturb.perturb_input = lambda x_, a, b, c: perturb.downscale(x_)
eng = get_engine("FE")(cfg, "Train")
eng.model._dec_dropout = False
if "nowarm" in mode:
    eng.warmup_step = 0
if "manual" in mode:
    # the loop of tests/test_engine_gpu.py on this engine's members
    from oracle import param_fill
    scaler = torch.amp.GradScaler("cuda", init_scale=2 ** 10)
    tgt = param_fill.make_labels(4).to(eng.device)
    for i in range(3):
        x = param_fill.make_input(4, 256, 50 + i).to(eng.device)
        eng.optimizer.zero_grad()
        r = eng.train_unidefense_model(x, tgt, 50 + i, scaler, 2, 2)
        if "keep" not in mode:
            r = {k: v.detach().float().cpu() for k, v in r.items()}
        print("step", i, float(r["total_loss"]))
else:
    print(eng.train())
