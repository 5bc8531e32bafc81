"""Concrete engine behind ``engine.get_engine`` (reference: engine/forgery_engine.py, ocim_engine.py,
uniattack_engine.py — three copies of one train loop around AbstractEngine.train_unidefense_model that differ in
their dataset / metric / wandb plumbing, which is out of scope here, SURVEY.md §8 b / §5).

What this mirrors of the reference (file:line of forgery_engine.py):
  * config layout: ``config['model']`` = model name + ctor kwargs (:141), ``config['config']`` = optimizer / scheduler /
    warmup_step / lambda_* / local_rank (:133-160), ``config['data']['train_batch_size']`` = samples per class and rank;
  * model -> SyncBN + data parallel (:142-146; here HipDataParallel over RCCL when a process group exists),
    timm weight-decay groups + optimizer (:149-154), scheduler (:156), the four loss criteria (:157-162);
  * train(): zero_grad, one [real...; fake...] batch per step from two sources, linear lr warm-up (:271-274),
    train_unidefense_model, loss / accuracy trackers reduced over ranks (:279-287), a log line every log_steps.
What replaces the dataset classes: ``config['data']['iterator']`` — a callable ``(step, batch, size, device) ->
(images_real, labels_real, images_fake, labels_fake)``; without one, seeded synthetic batches of the configured size
(there is no dataset in this image).  test(): accuracy / real-class scores over ``config['data']['test_iterator']``
(or synthetic batches) with the inference forward — the reference's ROC metrics stay out.  Checkpoints: the reference's
file names and format (engine/checkpoint.py); ``config['config']['dir']`` enables the save at the end of train(),
``resume`` the load at construction.
"""
import os

import torch
import torch.distributed as dist

from ..loss import get_loss
from ..model import load_model
from .abstract_engine import AbstractEngine
from .checkpoint import load_checkpoint, save_checkpoint
from .optim import build_optimizer, build_scheduler
from .parallel import wrap_data_parallel


def synthetic_batches(seed=0):
    """Default data source: x = 2*U(0,1)-1 faces, labels 0 = real, 1 = fake."""
    def it(step, batch, size, device):
        g = torch.Generator().manual_seed(seed * 1000003 + step)
        real = torch.rand(batch, 3, size, size, generator=g) * 2 - 1
        fake = torch.rand(batch, 3, size, size, generator=g) * 2 - 1
        return (real.to(device), torch.zeros(batch, dtype=torch.long, device=device),
                fake.to(device), torch.ones(batch, dtype=torch.long, device=device))
    return it


class TrainEngine(AbstractEngine):
    path = "engine/train_engine.py"

    def __init__(self, config, stage="Train"):
        super().__init__(config, stage)
        if not torch.cuda.is_available():
            raise RuntimeError("unidefense_amd engines run on an MI355X GPU only (no CPU path)")
        cfg = config["config"]
        self.local_rank = int(cfg.get("local_rank", os.environ.get("LOCAL_RANK", 0)))
        self.device = torch.device(f"cuda:{self.local_rank}")
        torch.cuda.set_device(self.device)
        if int(os.environ.get("WORLD_SIZE", "1")) > 1 and not dist.is_initialized():
            dist.init_process_group(cfg.get("distribute", {}).get("backend", "nccl"), device_id=self.device)
        model_cfg = dict(config["model"])
        self.model_name = model_cfg.pop("name")
        data_cfg = config.get("data", {})
        self.batch = int(data_cfg.get("train_batch_size", 16))
        self.size = int(data_cfg.get("size", 256))
        self.num_steps = int(cfg.get("num_steps", data_cfg.get("num_steps", 100)))
        self.log_steps = int(cfg.get("log_steps", data_cfg.get("log_steps", 50)))
        self.warmup_step = int(cfg.get("warmup_step", 0))
        self.train_iterator = data_cfg.get("iterator") or synthetic_batches(self.local_rank)
        self.test_iterator = data_cfg.get("test_iterator") or synthetic_batches(10007 + self.local_rank)

        self.best_step, self.best_auc, self.best_acc = 1, 0., 0.                    # forgery_engine.py:159-161
        self.model = load_model(self.model_name)(**model_cfg).to(self.device)
        # config.precision: "fp32" (default: fp32-accurate GEMMs, the reference's arithmetic) or "fp16" — BASELINE
        # configs[4]: fp16 MFMA operands with fp32 accumulation in every GEMM (ud_gemm path 3, process-wide) and, on the
        # EfficientNet model, fp16 storage of the MBConv trunk's activations; the GradScaler below is what keeps the half
        # gradients in range.  The reference has no such mode (autocast(enabled=False), abstract_engine.py:208).
        self.precision = str(cfg.get("precision", "fp32")).lower()
        if self.precision not in ("fp32", "fp16"):
            raise ValueError(f"config.precision must be 'fp32' or 'fp16', got {self.precision!r}")
        # The GEMM path is process-wide state of the library: an fp16 engine wants the fp16-MFMA kernel (3), an fp32 engine the
        # path in force when it is built (the library's UD_GEMM_PATH default or a caller's ud_gemm_set_path; never another engine's 3).  Building an engine does not touch it (an engine that still
        # lives keeps its arithmetic); each engine selects its own on entry to train() / test() / validate() / train_unidefense_model(), so two engines
        # of different precision in one process (an A/B run, a notebook) each compute in theirs.
        from .. import lib as _lib
        cur = _lib.call("ud_gemm_get_path")          # what the process (UD_GEMM_PATH at load) or a caller's ud_gemm_set_path chose
        self._gemm_path = 3 if self.precision == "fp16" else (cur if cur != 3 else 0)
        if self.precision == "fp16":
            self.model.half_storage = True
        self.model_without_ddp = self.model
        if dist.is_available() and dist.is_initialized():
            self.model = wrap_data_parallel(self.model, self.local_rank)           # SyncBN + gradient exchange
        if self.config["config"].get("resume"):
            # the test stage evaluates best_model.bin (forgery_engine.py:202), training resumes from latest_model.bin
            best = stage == "Test" and os.path.exists(self._ckpt_path(best=True))
            if os.path.exists(self._ckpt_path(best)):
                meta = self._load_ckpt(best=best, train=stage == "Train")
                self.best_step = int(meta.get("best_step", self.best_step) or self.best_step)
                self.best_auc = float(meta.get("best_auc", self.best_auc) or self.best_auc)
                self.best_acc = float(meta.get("best_acc", self.best_acc) or self.best_acc)
        if stage == "Train":
            self.base_lr = float(cfg["optimizer"]["lr"])
            self.optimizer = build_optimizer(self.model_without_ddp, cfg["optimizer"])
            self.scheduler = build_scheduler(self.optimizer, cfg.get("scheduler"))
            self.loss_criterion = {"softmax": get_loss("cross_entropy", self.device),
                                   "triplet": get_loss("aw_triplet", self.device),
                                   "kl_div": get_loss("kl_div", self.device),
                                   "fac": get_loss("factorization", self.device)}

    # ---- checkpoints in the reference's file names / format (forgery_engine.py:215-223) ------------------------------
    def _ckpt_path(self, best=False):
        return os.path.join(self.config["config"].get("dir", "."), "best_model.bin" if best else "latest_model.bin")

    def _any_rank(self, flag):
        """True on every rank iff `flag` is true on at least one (a no-op without a process group): decisions that guard a
        collective are taken from this, never from rank-local state such as the file system."""
        if not (dist.is_available() and dist.is_initialized()):
            return bool(flag)
        t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=self.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return bool(t.item())

    def _check_exchange(self):
        """A SyncBN peer exchange that timed out has poisoned this step's statistics (NaN): raise before anything is
        logged, validated or saved from them — on EVERY rank together, so that no rank is left waiting in the barrier /
        all_reduce that follows while another has already raised."""
        exchange = getattr(self.model, "bn_exchange", None)
        err = None
        if exchange is not None:
            try:
                exchange.check()
            except Exception as e:          # noqa: BLE001 — re-raised below, after the ranks have agreed
                err = e
        if self._any_rank(err is not None):
            raise err if err is not None else RuntimeError("SyncBN peer exchange failed on another rank")

    def _save_ckpt(self, step, best=False):
        self._check_exchange()
        if self.local_rank == 0:
            save_checkpoint(self.model_without_ddp, self._ckpt_path(best), step, self.best_step, self.best_auc,
                            self.best_acc)
        if dist.is_available() and dist.is_initialized():
            dist.barrier()              # the other ranks start the next step together with the writer (forgery_engine.py:219)

    def _load_ckpt(self, best=False, train=False):
        return load_checkpoint(self.model_without_ddp, self._ckpt_path(best))

    # ------------------------------------------------------------------------------------------------------------
    def _mean_over_ranks(self, values):
        t = torch.stack([v.detach().float().reshape(()) for v in values])
        if dist.is_available() and dist.is_initialized():
            dist.all_reduce(t)                      # one packed metric exchange per log step
            t /= dist.get_world_size()
        return t.tolist()

    def _select_gemm_path(self):
        from .. import lib as _lib
        if _lib.call("ud_gemm_get_path") != self._gemm_path:
            _lib.call("ud_gemm_set_path", self._gemm_path)

    def train(self):
        try:
            self._select_gemm_path()
            grad_scalar = torch.amp.GradScaler("cuda", init_scale=2 ** 10)        # forgery_engine.py:228
            sums, count, correct, seen, last = {}, 0, 0, 0, {}
            for cur_step in range(1, self.num_steps + 1):
                self.model.train()
                self.optimizer.zero_grad()
                xr, yr, xf, yf = self.train_iterator(cur_step, self.batch, self.size, self.device)
                in_data, in_tgt = torch.cat([xr, xf], 0).contiguous(), torch.cat([yr, yf], 0)
                if self.warmup_step != 0 and cur_step <= self.warmup_step:           # :271-274
                    for group in self.optimizer.param_groups:
                        group["lr"] = self.base_lr * float(cur_step) / self.warmup_step
                out = self.train_unidefense_model(in_data, in_tgt, cur_step, grad_scalar, yr.shape[0], yf.shape[0])
                for k, v in out.items():
                    if "loss" in k:
                        sums[k] = sums.get(k, 0.0) + v.detach()
                count += 1
                correct += (out["cls_out"].argmax(1) == in_tgt).sum()
                seen += in_tgt.numel()
                if cur_step % self.log_steps == 0 or cur_step == self.num_steps:
                    self._check_exchange()       # SyncBN peer exchange: a missing rank surfaces here, not as a silent NaN
                    keys = sorted(sums)
                    vals = self._mean_over_ranks([sums[k] / count for k in keys] + [correct / seen])
                    last = dict(zip(keys, vals[:-1]))
                    last["acc"], last["lr"], last["step"] = vals[-1], self.optimizer.param_groups[0]["lr"], cur_step
                    if self.local_rank == 0:
                        print("Train Iter (%d/%d), Loss %.4f, Triplet %.4f, Spat %.4f, Freq %.4f, ACC %.4f, LR %.6f" % (
                            cur_step, self.num_steps, last.get("total_loss", 0.0), last.get("triplet_loss", 0.0),
                            last.get("real_rec_loss", 0.0), last.get("real_freq_loss", 0.0), last["acc"], last["lr"]))
            if self.config["config"].get("dir"):
                # best_model.bin (what the reference's test stage reads, forgery_engine.py:202-207) is validate()'s to
                # choose, by validation AUC + ACC.  Only a run that never validated — no best file, no best record — gets
                # its final weights there, with the TRAIN accuracy of the last log window kept apart from the validation
                # record (best_auc / best_acc are validation numbers).
                self.last_train_acc = float(last.get("acc", 0.0))
                # (_save_ckpt ends in a barrier: the decision is rank 0's, broadcast — a rank that stats the directory a
                # moment after rank 0 created the file must not skip a barrier the others run)
                has_best = self._any_rank(self.local_rank == 0 and os.path.exists(self._ckpt_path(best=True)))
                if self.best_auc == 0.0 and self.best_acc == 0.0 and not has_best:
                    self.best_step = self.num_steps
                    self._save_ckpt(self.num_steps, best=True)
                self._save_ckpt(self.num_steps)
            return last
        except Exception:
            if dist.is_available() and dist.is_initialized():                          # :315-318
                dist.destroy_process_group()
            raise

    @torch.no_grad()
    def _score(self, batches):
        """p(real) = softmax(cls_out)[:, 0] of `batches` test batches (forgery_engine.py:349,437), gathered over the
        ranks with two device all_gathers (engine/metrics.py:gather_scores replaces dist.all_gather_object, :374-375)."""
        from .metrics import gather_scores
        self.model.eval()
        scores, labels = [], []
        for step in range(1, batches + 1):
            xr, yr, xf, yf = self.test_iterator(step, self.batch, self.size, self.device)
            out = self.model(torch.cat([xr, xf], 0).contiguous())          # inference forward: no tape, running statistics
            scores.append(torch.softmax(out["cls_out"], 1)[:, 0])
            labels.append(torch.cat([yr, yf], 0))
        return gather_scores(torch.cat(scores), torch.cat(labels))

    def test(self, batches=4):
        """The reference's test stage (forgery_engine.py:423-452): scores -> cal_metrics (EER, HTER = ACER, TPR@5 %, AUC,
        ACC, ...) over the frames of all ranks."""
        from .metrics import cal_metrics
        self._select_gemm_path()
        scores, labels = self._score(batches)
        sc, lb = scores.float().cpu().numpy(), labels.cpu().numpy()
        ret = {"scores": scores.cpu(), "labels": labels.cpu(), "acc": float(((sc < 0.5).astype(int) == lb).mean())}
        if len(set(lb.tolist())) == 2:                     # a ROC needs both classes
            ret.update(cal_metrics(lb, sc, threshold=0.5))
            if self.local_rank == 0:
                print("Test | EER %.4f, HTER %.4f, TPR 5%% %.4f, AUC %.4f, ACC %.4f" % (
                    ret["EER"], ret["ACER"], ret["TPR5%"], ret["AUC"], ret["ACC"]))
        return ret

    def validate(self, step, batches=4):
        """The reference's validate (forgery_engine.py:320-421) without the figure / wandb plumbing: metrics over all
        ranks, best-so-far record (AUC + ACC), best_model.bin / latest_model.bin on rank 0."""
        self._select_gemm_path()
        self._check_exchange()
        ret = self.test(batches)
        if "AUC" in ret and ret["AUC"] + ret["ACC"] > self.best_auc + self.best_acc:
            self.best_auc, self.best_acc, self.best_step = float(ret["AUC"]), float(ret["ACC"]), int(step)
            if self.config["config"].get("dir"):
                self._save_ckpt(step, best=True)
        if self.config["config"].get("dir"):
            self._save_ckpt(step, best=False)
        return ret
