"""GPU, informational: the full two-pass train step of the engine (AbstractEngine.train_unidefense_model: clean pass +
perturbed pass, two fused-AdamW steps) on UDEB4 256x256, the engine's default execution (each pass replayed from its
hipGraph; perturbation / optimizer / scheduler between them on the host's schedule).
usage: bench_train_step.py [batch=32] [steps=8]      UD_BENCH_SEED seeds the host RNG of the timed steps (A/B runs see the
same sequence of perturbation branches)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unidefense_amd.engine import AbstractEngine
from unidefense_amd.engine.optim import build_optimizer
from unidefense_amd.loss import LOSSES
from unidefense_amd.model import load_model
dev = torch.device("cuda:0")
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
m = load_model("UDEB4")(extractor="efficientnet-b4", num_classes=2, drop_rate=0.5).to(dev).train()
eng = AbstractEngine({"config": dict(lambda_triplet=0.1, lambda_recons=0.1, lambda_freq=1.0, lambda_mask=0.1, lambda_fac=0.1)})
eng.model, eng.device, eng.num_steps, eng.warmup_step = m, dev, 1000, 0
eng.optimizer = build_optimizer(m, dict(name="adamw", lr=1e-4, betas=[0.9, 0.999], weight_decay=5e-6, amsgrad=True))
eng.scheduler = torch.optim.lr_scheduler.StepLR(eng.optimizer, step_size=22500, gamma=0.5)
eng.loss_criterion = {"softmax": LOSSES["cross_entropy"], "triplet": LOSSES["aw_triplet"], "kl_div": LOSSES["kl_div"],
                      "fac": LOSSES["factorization"]}
x = (2 * torch.rand(bs, 3, 256, 256) - 1).to(dev)
tgt = torch.tensor([0] * (bs // 2) + [1] * (bs // 2), device=dev)
scaler = torch.amp.GradScaler("cuda", init_scale=2 ** 10, enabled=False)
def step(i):
    eng.optimizer.zero_grad()
    return eng.train_unidefense_model(x, tgt, 200 + i, scaler, bs // 2, bs // 2)
for i in range(3): step(i)
torch.manual_seed(int(os.environ.get("UD_BENCH_SEED", "0")))      # the same sequence of host-random perturbation branches in every run (A/B)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
for i in range(n): r = step(i)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print("two-pass train step: %.1f ms  ->  %.1f train images/s (each image goes through 2 passes); total_loss %.4f" % (dt * 1e3, bs / dt, float(r["total_loss"])))
