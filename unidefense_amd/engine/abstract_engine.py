"""AbstractEngine: the two-pass UniDefense train step on the HIP model.

Mirror of the reference's ``engine/abstract_engine.py`` for the hot path only: ``train_unidefense_model``
(:207-381) keeps its signature, return dict, loss weights and optimizer/scheduler/scaler call order, including
the reference's quirk that gradients are NOT zeroed between the two passes (``zero_grad`` is the caller's,
once per step: engine/forgery_engine.py:241).  Dataset plumbing, wandb, checkpoints and evaluation are out
of scope.
"""
import os
import random

import numpy as np
import torch
import torch.distributed as dist

from ..config import cfg


def _zero(device):
    return torch.tensor(0.0, device=device)


class AbstractEngine(object):
    path = "engine/abstract_engine.py"

    # Concrete engines (or tests) provide: model, optimizer, scheduler, loss_criterion
    # {"softmax","triplet","kl_div","fac"}, config, num_steps, warmup_step, device.
    def __init__(self, config=None, stage="Train"):
        feasible_stage = ["Train", "Test"]
        if stage not in feasible_stage:
            raise ValueError(f"stage should be in {feasible_stage}, but found '{stage}'")
        self.config = config or {}
        self.model = self.optimizer = self.scheduler = self.loss_criterion = None
        self.num_steps, self.warmup_step, self.device = 1, 0, None
        # hipGraph replay of the two passes (90 ms instead of 448 ms per train step at bs 32); cfg.engine_graph = False or
        # engine.use_graphs = False runs every launch eagerly
        self.use_graphs = cfg.engine_graph
        self._graphs = {}
        self.max_graph_sets = 4          # distinct (shape, split, kl) sets kept captured at a time

    @staticmethod
    def fixed_randomness(seed=42):
        random.seed(seed)
        np.random.seed(seed)
        torch.manual_seed(seed)
        if torch.cuda.is_available():
            torch.cuda.manual_seed_all(seed)

    def to_device(self, items):
        return [obj.to(self.device) for obj in items]

    # ------------------------------------------------------------------------------------------
    def _lam(self, key):
        return self.config["config"].get(key, 1.0)

    def _common_terms(self, out_dict, in_tgt, sum_real, sum_fake):
        """Loss terms both passes share (engine/abstract_engine.py:232-259 and :301-328)."""
        cls_out = out_dict["cls_out"]
        loss_dict = out_dict.get("loss_dict", dict())
        t = {}
        if loss_dict.get("triplet") is not None:
            crit = self.loss_criterion["triplet"]
            hint = hasattr(crit, "n_real")
            if hint:
                crit.n_real = sum_real                  # avoids a device read-back per feature
            try:
                t["triplet"] = sum(crit(feat, in_tgt) for feat in loss_dict["triplet"])
            finally:
                if hint:                                # the criterion object is shared (loss.LOSSES): no leak
                    crit.n_real = None
        else:
            t["triplet"] = _zero(self.device)
        for key, name in (("spatial", "rec"), ("freq", "freq")):
            if loss_dict.get(key) is not None:
                t["real_" + name] = torch.mean(loss_dict[key].narrow(0, 0, sum_real))
                t["fake_" + name] = torch.mean(loss_dict[key].narrow(0, sum_real, sum_fake))
            else:
                t["real_" + name] = t["fake_" + name] = _zero(self.device)
        if cls_out.shape[-1] == 1:
            t["cls"] = self.loss_criterion["softmax"](cls_out.squeeze(), in_tgt.float())
        else:
            t["cls"] = self.loss_criterion["softmax"](cls_out, in_tgt)
        return t

    def _backward(self, scaled_loss):
        """backward() of one pass.  The step's second backward adds onto the first's gradients with one multi-tensor launch
        instead of autograd's 504 AccumulateGrad launches (model/unidefense.py:_accumulate_in_place); the model's opt-in flag is
        set for the duration of THIS call only, so a user's own torch.autograd.grad / backward(inputs=...) through the model
        always takes autograd's road — and never under torch's DDP, whose reducer hooks hang on the AccumulateGrad nodes."""
        core = getattr(self.model, "module", self.model)
        own = hasattr(core, "_run") and not isinstance(self.model, torch.nn.parallel.DistributedDataParallel)
        if own:
            core._ud_inplace_accumulate = True
        try:
            scaled_loss.backward()
        finally:
            if own:
                core._ud_inplace_accumulate = False

    def _backward_and_step(self, total_loss, grad_scalar):
        self._backward(grad_scalar.scale(total_loss))
        grad_scalar.step(self.optimizer)
        grad_scalar.update()

    @staticmethod
    def _barrier():
        if dist.is_available() and dist.is_initialized():
            dist.barrier()

    # ---- the two passes, shared by the eager and the graph-captured step ---------------------------------------
    def _pass1(self, in_data, in_tgt, sum_real, sum_fake):
        """Clean pass (engine/abstract_engine.py:210-281): returns (ret_dict, targets for pass 2, total_loss)."""
        out_dict = self.model(in_data)
        loss_dict = out_dict.get("loss_dict", dict())
        has_fm = loss_dict.get("freq_mask") is not None
        has_sm = loss_dict.get("spat_mask") is not None
        gts = {"freq_mask": loss_dict["freq_mask"].clone().detach() if has_fm else None,
               "spat_mask": loss_dict["spat_mask"].clone().detach() if has_sm else None,
               "fac": loss_dict["factorization"].clone().detach()}
        # the scalar tail (criteria, means, weighted sum) as two HIP launches where the criteria are the standard ones
        # (loss/pass_tail.py); the torch formulation below stays the definition and the fallback
        f = self._fused_tail(out_dict, in_tgt, sum_real, sum_fake,
                             dict(cls=1.0, mask=self._lam("lambda_mask"), triplet=self._lam("lambda_triplet"),
                                  rec=self._lam("lambda_recons"), freq=self._lam("lambda_freq")))
        if f is not None:
            t = {"cls": f["cls"], "triplet": f["triplet"], "real_rec": f["real_rec"], "fake_rec": f["fake_rec"],
                 "real_freq": f["real_freq"], "fake_freq": f["fake_freq"]}
            total_loss = f["total"]
        else:
            freq_mask_loss = torch.mean(loss_dict["freq_mask"]) if has_fm else _zero(self.device)
            spat_mask_loss = torch.mean(loss_dict["spat_mask"]) if has_sm else _zero(self.device)
            t = self._common_terms(out_dict, in_tgt, sum_real, sum_fake)
            total_loss = t["cls"] + self._lam("lambda_mask") * freq_mask_loss + self._lam("lambda_mask") * spat_mask_loss \
                + self._lam("lambda_triplet") * t["triplet"] + self._lam("lambda_recons") * t["real_rec"] \
                + self._lam("lambda_freq") * t["real_freq"]
        ret_dict = {
            "total_loss": total_loss, "cls_out": out_dict["cls_out"], "cls_loss": t["cls"],
            "triplet_loss": t["triplet"], "real_rec_loss": t["real_rec"], "fake_rec_loss": t["fake_rec"],
            "real_freq_loss": t["real_freq"], "fake_freq_loss": t["fake_freq"],
        }
        return ret_dict, gts, total_loss

    def _kld(self, pred, gt):
        """mask alignment: KL between the log-softmaxed flattened masks of the two passes (abstract_engine.py:296-300)"""
        pred = torch.log_softmax(pred.reshape(pred.shape[0], -1), dim=-1)
        gt = torch.log_softmax(gt.reshape(gt.shape[0], -1), dim=-1)
        return self.loss_criterion["kl_div"](pred, gt)

    def _fused_tail(self, out_dict, in_tgt, sum_real, sum_fake, weights, masks=True):
        if not getattr(self, "fused_loss_tail", True):
            return None
        from ..loss.pass_tail import pass_tail
        return pass_tail(out_dict, in_tgt, sum_real, sum_fake, self.loss_criterion, weights, masks=masks)

    def _pass2(self, out_dict, in_tgt, sum_real, sum_fake, gts, kl):
        """Loss assembly of the perturbed pass (engine/abstract_engine.py:294-371); kl = cur_step > 0.1 num_steps."""
        loss_dict = out_dict.get("loss_dict", dict())
        has_fm, has_sm = gts["freq_mask"] is not None, gts["spat_mask"] is not None
        f = self._fused_tail(out_dict, in_tgt, sum_real, sum_fake,
                             dict(cls=0.1, mask=self._lam("lambda_mask"), triplet=self._lam("lambda_triplet"),
                                  rec=0.1 * self._lam("lambda_recons"), freq=0.1 * self._lam("lambda_freq")), masks=not kl)
        if f is not None:
            fac_loss = self.loss_criterion["fac"](loss_dict["factorization"], gts["fac"])
            total_loss = f["total"] + self._lam("lambda_fac") * fac_loss
            if kl:
                freq_mask_loss = self._kld(loss_dict["freq_mask"], gts["freq_mask"]) if has_fm else torch.zeros_like(fac_loss)
                spat_mask_loss = self._kld(loss_dict["spat_mask"], gts["spat_mask"]) if has_sm else torch.zeros_like(fac_loss)
                total_loss = total_loss + self._lam("lambda_mask") * freq_mask_loss + self._lam("lambda_mask") * spat_mask_loss
            else:
                freq_mask_loss, spat_mask_loss = f["freq_mask"], f["spat_mask"]
            return {"freq_mask_loss": freq_mask_loss, "spat_mask_loss": spat_mask_loss, "fac_loss": fac_loss}, total_loss
        t = self._common_terms(out_dict, in_tgt, sum_real, sum_fake)
        zero_like = torch.zeros_like(t["cls"])
        if kl:
            # mask alignment: KL between the log-softmaxed flattened masks of the two passes
            def kld(pred, gt):
                pred = torch.log_softmax(pred.reshape(pred.shape[0], -1), dim=-1)
                gt = torch.log_softmax(gt.reshape(gt.shape[0], -1), dim=-1)
                return self.loss_criterion["kl_div"](pred, gt)
            freq_mask_loss = kld(loss_dict["freq_mask"], gts["freq_mask"]) if has_fm else zero_like
            spat_mask_loss = kld(loss_dict["spat_mask"], gts["spat_mask"]) if has_sm else zero_like
        else:
            freq_mask_loss = torch.mean(loss_dict["freq_mask"]) if has_fm else zero_like
            spat_mask_loss = torch.mean(loss_dict["spat_mask"]) if has_sm else zero_like
        fac_loss = self.loss_criterion["fac"](loss_dict["factorization"], gts["fac"])
        total_loss = 0.1 * t["cls"] + self._lam("lambda_mask") * freq_mask_loss \
            + self._lam("lambda_mask") * spat_mask_loss + self._lam("lambda_triplet") * t["triplet"] \
            + self._lam("lambda_recons") * 0.1 * t["real_rec"] + self._lam("lambda_freq") * 0.1 * t["real_freq"] \
            + self._lam("lambda_fac") * fac_loss
        return {"freq_mask_loss": freq_mask_loss, "spat_mask_loss": spat_mask_loss, "fac_loss": fac_loss}, total_loss

    def train_unidefense_model(self, in_data, in_tgt, cur_step, grad_scalar, sum_real=None, sum_fake=None):
        """One train step = clean pass + perturbed/consistency pass, each with its own backward and optimizer
        step.  Batch order must be [real...; fake...].
        With ``self.use_graphs`` (env UD_ENGINE_GRAPH=1) the forward + loss + backward of each pass is captured into a
        hipGraph on the second call with a given shape and replayed afterwards; perturbation, optimizer, scaler and
        scheduler calls stay outside the graphs (host-side randomness / synchronisation)."""
        kl = cur_step > self.num_steps * 0.1
        if hasattr(self, "_select_gemm_path"):
            self._select_gemm_path()          # this engine's arithmetic, also when the step is called directly
        if self.use_graphs and not getattr(self.model, "rng_queue", None) and in_data.is_cuda:
            return self._train_graphed(in_data, in_tgt, cur_step, grad_scalar, sum_real, sum_fake, kl)
        # ---------------- pass 1: clean input ------------------------------------------------------
        ret_dict, gts, total_loss = self._pass1(in_data, in_tgt, sum_real, sum_fake)
        self._backward_and_step(total_loss, grad_scalar)
        self._barrier()
        # ---------------- pass 2: perturbed input + consistency with pass 1 -----------------------
        pert_real_list = torch.arange(sum_real)[torch.randperm(sum_real)]
        pert_fake_list = torch.arange(sum_fake)[torch.randperm(sum_fake)]
        out_dict = self.model(in_data, pert_real_list=pert_real_list, pert_fake_list=pert_fake_list,
                              preserve_color=True)
        ret2, total_loss = self._pass2(out_dict, in_tgt, sum_real, sum_fake, gts, kl)
        ret_dict.update(ret2)
        self._backward_and_step(total_loss, grad_scalar)
        if self.warmup_step == 0 or cur_step > self.warmup_step:
            self.scheduler.step()
        self._barrier()
        # detached: a caller that keeps the dict must not keep this step's autograd graph (and its AccumulateGrad nodes,
        # bound to this stream) alive — the next step may be captured into a hipGraph on another stream, and a backward
        # that meets those stale nodes makes the capture depend on a non-capturing stream (crash at capture end)
        return {k: v.detach() for k, v in ret_dict.items()}

    # ---- graph-captured step ------------------------------------------------------------------------------------
    def _perturbed(self, in_data, sum_real, sum_fake):
        from ..model import perturb
        pert_real_list = torch.arange(sum_real)[torch.randperm(sum_real)]
        pert_fake_list = torch.arange(sum_fake)[torch.randperm(sum_fake)]
        with torch.no_grad():
            return perturb.perturb_input(in_data, pert_real_list, pert_fake_list, True).contiguous().to(torch.float32)

    def _train_graphed(self, in_data, in_tgt, cur_step, grad_scalar, sum_real, sum_fake, kl):
        # Everything a capture bakes in is part of the key: shapes, the real/fake split, the KL branch, the loss
        # weights, and the GradScaler whose scale tensor `grad_scalar.scale(total)` reads inside the graph (a fresh
        # scaler per train() call must not replay graphs that multiply by the previous scaler's buffer).
        lam = tuple(self._lam(k) for k in ("lambda_mask", "lambda_triplet", "lambda_recons", "lambda_freq", "lambda_fac"))
        key = (tuple(in_data.shape), int(sum_real), int(sum_fake), bool(kl), id(grad_scalar), lam)
        if key not in self._graphs:
            # evict sets that can no longer be hit: other scalers / loss weights, and the kl=False twin once the step
            # counter passed the threshold (each set pins a private memory pool with a full step of activations)
            for k_ in [k_ for k_ in self._graphs if k_[4:] != key[4:] or (k_[:3] == key[:3] and kl and not k_[3])]:
                del self._graphs[k_]
            while len(self._graphs) >= self.max_graph_sets:
                del self._graphs[next(iter(self._graphs))]
        st = self._graphs.setdefault(key, {"calls": 0, "scaler": grad_scalar})     # keeps the scaler's id unique
        st["calls"] += 1
        params = [p for p in self.model.parameters() if p.requires_grad]
        if st["calls"] == 1:
            # eager first call: warms up every lazily created buffer (scaler state, workspaces, optimizer state)
            saved, self.use_graphs = self.use_graphs, False
            try:
                return self.train_unidefense_model(in_data, in_tgt, cur_step, grad_scalar, sum_real, sum_fake)
            finally:
                self.use_graphs = saved
        mode = "thread_local" if (dist.is_available() and dist.is_initialized()) else "global"
        if "g1" not in st:
            st["x"], st["tgt"] = in_data.clone(), in_tgt.clone()
            st["noise"] = torch.empty_like(st["x"])
            for p in params:
                p.grad = None
            torch.cuda.synchronize()
            st["g1"] = torch.cuda.CUDAGraph()
            with torch.cuda.graph(st["g1"], capture_error_mode=mode):
                ret1, gts, total1 = self._pass1(st["x"], st["tgt"], sum_real, sum_fake)
                self._backward(grad_scalar.scale(total1))
            st["ret1"], st["gts"] = ret1, gts
            st["grads"] = [p.grad for p in params]          # static buffers the replays write
            st["pool"] = st["g1"].pool()
        st["x"].copy_(in_data)
        st["tgt"].copy_(in_tgt)
        st["g1"].replay()
        for p, g in zip(params, st["grads"]):
            p.grad = g
        grad_scalar.step(self.optimizer)
        grad_scalar.update()
        self._barrier()
        st["noise"].copy_(self._perturbed(in_data, sum_real, sum_fake))
        if "g2" not in st:
            torch.cuda.synchronize()
            st["g2"] = torch.cuda.CUDAGraph()
            with torch.cuda.graph(st["g2"], pool=st["pool"], capture_error_mode=mode):
                out_dict = self.model(st["x"], noise_x=st["noise"])
                ret2, total2 = self._pass2(out_dict, st["tgt"], sum_real, sum_fake, st["gts"], kl)
                self._backward(grad_scalar.scale(total2))  # accumulates in place onto the pass-1 gradient buffers
            st["ret2"] = ret2
        st["g2"].replay()
        grad_scalar.step(self.optimizer)
        grad_scalar.update()
        if self.warmup_step == 0 or cur_step > self.warmup_step:
            self.scheduler.step()
        self._barrier()
        ret = {k: v.detach().clone() for k, v in st["ret1"].items()}
        ret.update({k: v.detach().clone() for k, v in st["ret2"].items()})
        return ret
