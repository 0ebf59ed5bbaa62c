"""Hulc2 — the low-level policy LightningModule on MI355X kernels.

Mirrors hulc2.models.hulc2.Hulc2 (reference hulc2/models/hulc2.py:27-508) for the training path: the same 16
constructor arguments (un-instantiated configs, `_recursive_: false`), `setup_input_sizes`, `training_step`,
`lmp_train`, `compute_kl_loss`, `clip_auxiliary_loss`, `configure_optimizers`, `set_kl_beta`, the same logged
metric names and the same attribute names (hence state_dict keys); and for validation / rollout inference
(SURVEY.md §8 row f-1): `lmp_val`, `validation_step`, `reset`, `step`, `predict_with_plan`, `get_pp_plan_vision/_lang`.

Differences that do not change results:
  * the categorical plan sample can be injected through `dataset_batch["plan_idx"]` (parity tests); otherwise it
    is drawn on-device by the counter RNG (the reference uses torch.multinomial — no shared RNG stream exists)
  * `use_for_aux_lang_loss` is applied inside the CLIP loss kernel instead of boolean-mask indexing, so the step
    has no host synchronisation (the reference's `torch.any` / dynamic shapes, hulc2.py:391-394,490-493)
"""
import logging
from typing import Dict, Optional, Tuple

import numpy as np
import os

import torch
import torch.nn as nn

from hulc2_amd import functional as HF
from hulc2_amd import kernels as kn
from hulc2_amd.compat import LightningModule, instantiate
from hulc2_amd.utils.distributions import State

logger = logging.getLogger(__name__)


def _kernel_precision(fn):
    """Lightning (`precision: 16`, conf/trainer/play_trainer.yaml:3) calls the hooks inside `torch.autocast(fp16)`.  The kernels behind
    this module choose their own arithmetic (kernels.set_compute) and read their operands through raw pointers, so a framework op that
    autocast demoted to half would hand a kernel the wrong bytes: the hooks run with autocast off, as the reference itself does around
    `world_to_tcp_frame` (gripper_control.py:17).  A GradScaler's power-of-two loss scale passes through every backward kernel exactly."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *args, **kwargs):
        if torch.is_autocast_enabled():
            with torch.autocast(device_type="cuda", enabled=False):
                return fn(self, *args, **kwargs)
        return fn(self, *args, **kwargs)
    return wrapped


class Hulc2(LightningModule):
    def __init__(self, perceptual_encoder, plan_proposal, plan_recognition, language_encoder, language_goal, visual_goal,
                 action_decoder, kl_beta: float, kl_balancing_mix: float, optimizer, lr_scheduler, distribution,
                 use_clip_auxiliary_loss: bool, clip_auxiliary_loss_beta: float, replan_freq: int = 30, proj_vis_lang=None):
        super().__init__()
        self.perceptual_encoder = instantiate(perceptual_encoder, device=self.device)
        self.setup_input_sizes(self.perceptual_encoder, plan_proposal, plan_recognition, visual_goal, action_decoder, distribution)
        self.dist = instantiate(distribution)
        self.plan_proposal = instantiate(plan_proposal, dist=self.dist)
        self.plan_recognition = instantiate(plan_recognition, dist=self.dist)
        self.visual_goal = instantiate(visual_goal)
        self.lang_encoder = instantiate(language_encoder) if language_encoder else None
        self.language_goal = instantiate(language_goal, lang_net=self.lang_encoder) if language_goal else None
        self.action_decoder = instantiate(action_decoder)
        self.use_clip_auxiliary_loss = use_clip_auxiliary_loss
        self.clip_auxiliary_loss_beta = clip_auxiliary_loss_beta
        if use_clip_auxiliary_loss:
            self.logit_scale = nn.Parameter(torch.ones([]) * np.log(1 / 0.07))
            self.proj_vis_lang = instantiate(proj_vis_lang)
        self.kl_beta = kl_beta
        self.kl_balancing_mix = kl_balancing_mix
        self.modality_scope = "vis"
        self.optimizer_config = optimizer
        self.lr_scheduler = lr_scheduler
        self.save_hyperparameters()
        self.rollout_step_counter = 0
        self.replan_freq = replan_freq
        self.latent_goal = None
        self.plan = None

    @staticmethod
    def setup_input_sizes(perceptual_encoder, plan_proposal, plan_recognition, visual_goal, action_decoder, distribution):
        """hulc2.py:126-158: patch the `???` sizes of the child configs."""
        n = perceptual_encoder.latent_size
        plan_proposal.perceptual_features = n
        plan_recognition.in_features = n
        visual_goal.in_features = n
        action_decoder.perceptual_features = n
        if distribution.dist == "discrete":
            pf = distribution.class_size * distribution.category_size
        else:
            pf = distribution.plan_features
        plan_proposal.plan_features = plan_recognition.plan_features = action_decoder.plan_features = pf

    def configure_optimizers(self):
        """hulc2.py:185-198 (Adam lr 2e-4 + constant schedule).  The native trainer (hulc2_amd/trainer.py) replaces this
        with the fused arena Adam; under Lightning any torch optimizer works (the keeper re-derives the weight copies)."""
        cfg = self.optimizer_config
        tgt = cfg.get("_target_") if hasattr(cfg, "get") else None
        if tgt == "torch.optim.Adam" and not os.environ.get("HULC_TORCH_ADAM"):
            # round 6: the unchanged conf/model/optimizer/adam.yaml gets the drop-in SUBCLASS (hulc2_amd/optim.py): same update rule, same
            # hyper-parameters, same state_dict layout (checkpoints interchange), isinstance(opt, torch.optim.Adam) holds; its step is one
            # launch of the arena kernel on the step node's gradient arena and takes a GradScaler's device scalars without a host
            # synchronisation; any configuration it does not cover (amsgrad, several groups, CPU, ...) runs torch.optim.Adam.step() on the
            # same state.  HULC_TORCH_ADAM=1 hands out torch's own class.
            from ..optim import Adam
            opt = Adam(self.parameters(), **{k: v for k, v in cfg.items() if not str(k).startswith("_")})
        else:
            opt = instantiate(cfg, params=self.parameters())
        sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda _: 1.0)
        return {"optimizer": opt, "lr_scheduler": {"scheduler": sched, "interval": "step", "frequency": 1}}

    # ---- hot path ----------------------------------------------------------------------------------
    @staticmethod
    def _goal_site(is_lang: bool):
        """precision scope of a goal encoder: only the LANGUAGE goal is upstream of the contrastive head — in a bf16 step the
        site "goal" makes that encoder's forward exact and leaves the visual one on its bf16 launches; 'mixed' / 'fp32' steps treat both alike"""
        import contextlib
        if is_lang or kn.base_mode() != "bf16":
            return kn.site_scope("goal")
        return contextlib.nullcontext()

    def lmp_train(self, perceptual_emb, latent_goal, train_acts, robot_obs, plan_idx: Optional[torch.Tensor] = None
                  ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, State, State, torch.Tensor]:
        """hulc2.py:200-245; returns the prior/posterior *states* in place of torch.distributions objects."""
        pp_state = self.plan_proposal(perceptual_emb[:, 0], latent_goal)
        pr_state, seq_feat = self.plan_recognition(perceptual_emb)
        site = 0xA11CE if "lang" in self.modality_scope else 0xB0B       # distinct RNG sites for the two modalities
        sampled_plan, _ = self.dist.rsample_plan(pr_state, seed=site, idx=plan_idx)
        action_loss = self.action_decoder.loss(sampled_plan, perceptual_emb, latent_goal, train_acts, robot_obs)
        kl_loss = self.compute_kl_loss(pp_state, pr_state)
        return kl_loss, action_loss, action_loss + kl_loss, pp_state, pr_state, seq_feat

    def _keep_weight_copies_fresh(self):
        """Under an EXTERNAL optimizer (Lightning + torch.optim.Adam, hulc2/training.py:79-82) nobody maintains the kernel-side copies of the
        weights (bf16 shadows, transposed tiles, packed fragments, split-operand remainders, conv repacks): shadow.weight_operand re-derives
        each of them per parameter and layout with torch ops when the parameter's version has changed — ~200 small launches per step.  The
        first training-mode step on a GPU without an ArenaTrainer therefore installs ArenaTrainer(shadows_only=True): the parameters move into
        one arena (same Parameter objects, same values: optimizers, DDP and checkpoints are unaffected) and all copies are re-made by five
        launches whenever the optimizer has stepped.  bf16 arithmetic modes only; HULC_NO_AUTO_SHADOWS=1 keeps the lazy per-parameter path.
        Round 5: the keeper also owns the gradient arena of the step node (hulc2_amd/stepnode.py; HULC_NO_STEP_NODE=1: weight copies only).
        Returns the keeper when it is this model's (else None: a full ArenaTrainer drives the step, or the lazy path is selected)."""
        if kn.base_mode() == "fp32" or os.environ.get("HULC_NO_AUTO_SHADOWS"):
            return None
        ref = self.__dict__.get("_hulc_arena_trainer")
        tr = ref() if ref is not None else None
        if tr is not None and tr.model is not self:               # (a deep copy of a model carries the original's weak reference along)
            tr = None
        if tr is None:
            from ..trainer import ArenaTrainer
            tr = self.__dict__["_hulc_shadow_keeper"] = ArenaTrainer(self, shadows_only=True,     # the model owns its keeper (a deepcopy of the
                                                                     step_node=not os.environ.get("HULC_NO_STEP_NODE"))    # model gets None here and builds its own)
        if getattr(tr, "shadows_only", False):
            tr.refresh_if_stale()
            return tr
        return None

    def _fresh_weight_copies(self) -> None:
        """every forward entry point that is NOT a training step (validation, rollout): under an external optimizer the keeper's copies are
        re-derived when a parameter has been written since (optimizer.step(), an EMA swap, weight surgery) — the arena views the kernels
        read carry no version of their own"""
        tr = self.__dict__.get("_hulc_shadow_keeper")
        if tr is not None and getattr(tr, "shadows_only", False):
            ref = self.__dict__.get("_hulc_arena_trainer")
            if ref is not None and ref() is tr:
                tr.refresh_if_stale()

    def _step_node(self, keeper):
        """the whole training step as one autograd node (hulc2_amd/stepnode.py) when this model's keeper owns a gradient arena and the call can
        take it: gradients enabled, no gradient being accumulated over several calls, the keeper's parameter set still the trainable one"""
        if keeper is None or not getattr(keeper, "step_node", False) or not torch.is_grad_enabled():
            return None
        node = self.__dict__.get("_hulc_step_node")
        if node is None or node.keeper is not keeper:
            from ..stepnode import StepNode
            node = self.__dict__["_hulc_step_node"] = StepNode(self, keeper)
        return node if node.usable() else None

    @_kernel_precision
    def training_step(self, batch: Dict[str, Dict], batch_idx: int) -> torch.Tensor:
        """hulc2.py:336-442."""
        node = None
        if self.training:
            # every training-mode call draws fresh dropout masks and a fresh latent-plan sample (the device RNG word, kernels.step_state),
            # under any trainer: Lightning + a torch optimizer never touches that word, ArenaTrainer marks it fresh for this step itself
            p0 = next(self.parameters())
            if p0.is_cuda:
                node = self._step_node(self._keep_weight_copies_fresh())
                if node is None:                               # (the step node walks the word itself: inside its captured forward graph)
                    kn.ensure_fresh_rng(p0.device)
        else:
            self._fresh_weight_copies()
        # under an external optimizer (Lightning's loop, hulc2/training.py:79-82) the step is ONE autograd node — eager at first, two replayed
        # hipGraphs from the third step of a configuration on (hulc2_amd/stepnode.py); otherwise the Functions of functional.py hang on the loss
        total_loss, logs = node(batch, batch_idx) if node is not None else self._training_step_impl(batch, batch_idx)
        for name, value, kw in logs:
            self.log(name, value, **kw)
        return total_loss

    def _training_step_impl(self, batch: Dict[str, Dict], batch_idx: int):
        """the arithmetic of training_step: -> (total loss, [(logged name, value, self.log keyword arguments)])"""
        kl_loss = action_loss = total_loss = lang_clip_loss = None
        clip_forked = False
        batch_size: Dict[str, int] = {}
        total_bs = 0

        def acc(a, b):
            return b if a is None else a + b

        # encoders, goal, prior/posterior, latent plan sample, KL, decoder (hulc2.py:380-386,228-242).  Modalities of identical
        # shape are stacked on the batch axis through every shared network (same per-row arithmetic, half the launches); only
        # the goal encoders differ per modality, and every loss is still the mean over its own modality's rows.
        per = []
        kl_stacked = None
        mods = list(batch.items())
        if self._batchable(mods):
            with kn.site_scope("enc"):
                emb_all = self.perceptual_encoder([db["rgb_obs"] for _, db in mods], None, None)
            if emb_all.requires_grad and emb_all.is_cuda:
                # when backward reaches the embedding every weight gradient outside the camera encoders is queued: HULC_WGRAD_EARLY=1 issues the
                # grouped launch on a second stream at that point (kernels.wgrad_flush_early; measured slower, off by default)
                emb_all.register_hook(lambda g_: kn.wgrad_flush_early(g_.device))
            B = mods[0][1]["actions"].shape[0]
            # the embedding's four consumers get their views from one fan-out node (one gather launch forward, ONE merge launch backward
            # instead of autograd's select / slice backward fills and three accumulate adds): emb[:, 0] -> prior, emb[:B, -1] of a leading
            # vision modality -> visual goal encoder, emb -> posterior, emb[..., lo:hi] time-major -> action decoder
            vis_first = "lang" not in mods[0][0] and all("lang" in sc for sc, _ in mods[1:])
            lo, hi = self.action_decoder.perceptual_emb_slice
            fan = vis_first and emb_all.is_cuda
            goal_all = None
            # round 6: the posterior (transformer trunk + its head) does not depend on the goal encoders and the prior (hulc2.py:228-233): it runs as
            # a second branch — a side stream eagerly, a branch of the captured graph in replay; autograd runs each branch's backward on the
            # stream of its forward, so the backward forks the same way.  Both branches are cooperative launches that own a CU per workgroup:
            # each keeps to half of the device (kernels.coop_share_scope, include/hulc2_amd.h hulc_set_coop_share).  HULC_FORK=0: one after the other.
            fork = fan and kn.fork_branches()
            if fan:
                emb0, emb_last, emb_rec, emb_dec_t = HF.EmbFanoutFn.apply(emb_all, B, lo, hi)
            if fork:
                cur, side = torch.cuda.current_stream(emb_all.device), kn.branch_stream(emb_all.device)
                side.wait_stream(cur)
                with torch.cuda.stream(side), kn.coop_share_scope(2):
                    pr_all, seq_all = self.plan_recognition(emb_rec)
            with kn.coop_share_scope(2 if fork else 1):           # (the prior branch's chain launches: half of the device next to the posterior's)
                if fan:
                    # selective precision: of the goal encoders only the LANGUAGE one is upstream of the contrastive head; inside a bf16
                    # step with site "goal" it alone runs its forward exactly, the visual one stays on its bf16 chain launch
                    lang_only_exact = kn.base_mode() == "bf16" and kn.get_compute() == "bf16" and "goal" in kn.fp32_sites() and len(mods) == 2
                    if lang_only_exact:
                        # both encoders in ONE launch, the language one from split operands (three MFMAs per product: fp32-class values) ...
                        pair = HF.dual_mlp(emb_last, self.visual_goal.mlp_layers(), self.language_goal.embed(mods[1][1]["lang"]),
                                           self.language_goal.mlp_layers(), exact_b=True)
                        if pair is None:        # ... or, where the pair does not fit the launch, on the exact-fp32 GEMMs
                            pre_v = self.visual_goal(emb_last, pre_ln=True)
                            with kn.site_scope("goal"):
                                pre_l = self.language_goal(mods[1][1]["lang"], pre_ln=True)
                            pair = (pre_v, pre_l)
                        pre = list(pair)
                    with kn.site_scope("goal"):
                        # the modalities' goal encoders stop in front of their LayerNorms, which then write the rows of the stacked goal tensor
                        # directly (no concatenation, no strided gradient slices on the way back)
                        if lang_only_exact:
                            pass
                        elif len(mods) == 2:    # the two goal MLPs (same hidden widths, their own weights) as one launch each way
                            pre = list(HF.dual_mlp(emb_last, self.visual_goal.mlp_layers(), self.language_goal.embed(mods[1][1]["lang"]),
                                                   self.language_goal.mlp_layers()))
                        else:
                            pre = [self.language_goal(db["lang"], pre_ln=True) if "lang" in scope else self.visual_goal(emb_last, pre_ln=True) for scope, db in mods]
                        goal_all = HF.layer_norm_cat(pre, [self.language_goal.ln if "lang" in scope else self.visual_goal.ln for scope, _ in mods], dim=0)
                    goals = [None] * len(mods)
                else:
                    embs = [emb_all[i * B:(i + 1) * B] for i in range(len(mods))]
                    emb0, emb_rec = emb_all[:, 0], emb_all
                    goals = []
                    for i, (scope, db) in enumerate(mods):
                        with self._goal_site("lang" in scope):
                            goals.append(self.language_goal(db["lang"]) if "lang" in scope else self.visual_goal(embs[i][:, -1]))
                if goal_all is None:
                    goal_all = torch.cat(goals, dim=0)
                with kn.site_scope("prior"):
                    pp_all = self.plan_proposal(emb0, goal_all)
            if fork:
                cur.wait_stream(side)                      # join: sample + KL consume both branches
                for t_ in (pr_all.logit, seq_all):
                    t_.record_stream(cur)
            else:
                pr_all, seq_all = self.plan_recognition(emb_rec)
            # the contrastive head sees the stacked rows when the language modality is the last segment: rows below row0 are masked out inside the
            # loss kernel (their gradient is exactly zero) — no slice of the pooled features / goals, no gradient scatter on the way back
            lang_ix = [i for i, (sc, _) in enumerate(mods) if "lang" in sc]
            stacked_head = fan and lang_ix == [len(mods) - 1] and self.use_clip_auxiliary_loss and len(mods) * B <= 128
            for i, (self.modality_scope, db) in enumerate(mods):
                if stacked_head and i == lang_ix[0]:
                    per.append((self.modality_scope, db, None, goal_all, seq_all, i * B, None))
                else:
                    per.append((self.modality_scope, db, None, goals[i] if goals[i] is not None else goal_all[i * B:(i + 1) * B],
                                seq_all[i * B:(i + 1) * B], None, None))
            # round 6: the contrastive head (two projections + the loss, forward and backward ~0.15 ms of small launches) needs the pooled
            # posterior features and the goals only — it runs as a branch beside sample / KL / the decoder's input projections, and its backward
            # beside the start of the decoder's.  The recurrent sweeps take every CU while they run, so nothing in this branch may be a
            # cooperative launch (coop_share_scope(0): the projections' chains fall back to GEMMs — a chain spinning on some CUs while the sweep
            # waits for all of them would never finish).
            if fork and self.use_clip_auxiliary_loss and any("lang" in p_[0] for p_ in per) and not os.environ.get("HULC_NO_CLIP_FORK"):
                side.wait_stream(cur)
                with torch.cuda.stream(side), kn.coop_share_scope(0):
                    for (scope_, db_, _e, goal_, seqf_, plan_, _k) in per:
                        if "lang" in scope_:
                            lang_clip_loss = acc(lang_clip_loss, self.clip_auxiliary_loss(seqf_, goal_, db_["use_for_aux_lang_loss"],
                                                                                          row0=plan_ if isinstance(plan_, int) else 0))
                clip_forked = True
            # sample, KL and decoder once over the stacked rows; the KL / decoder kernels return one mean per modality.  Sample + KL are ONE
            # autograd node (both consume the posterior's logits: their gradients meet inside the kernels, not in a fan-in add)
            idxs = [db.get("plan_idx") for _, db in mods]
            idx_all = torch.cat(idxs, dim=0) if all(i is not None for i in idxs) else None
            plan_all, _, kls = self.dist.rsample_plan_and_kl(pp_all, pr_all, 0xA11CE, idx_all, self.kl_beta, self.kl_balancing_mix, len(mods))
            kl_stacked = kls
            act_losses = self.action_decoder.loss_stacked(plan_all, emb_dec_t if fan else emb_all, goal_all,
                                                          [db["actions"] for _, db in mods], [db["state_info"]["robot_obs"] for _, db in mods],
                                                          len(mods), emb_tm=fan)
        else:
            for self.modality_scope, db in mods:
                with kn.site_scope("enc"):
                    emb = self.perceptual_encoder(db["rgb_obs"], db["depth_obs"], db["robot_obs"])
                with self._goal_site("lang" in self.modality_scope):
                    latent_goal = self.language_goal(db["lang"]) if "lang" in self.modality_scope else self.visual_goal(emb[:, -1])
                with kn.site_scope("prior"):
                    pp_state = self.plan_proposal(emb[:, 0], latent_goal)
                pr_state, seq_feat = self.plan_recognition(emb)
                site = 0xA11CE if "lang" in self.modality_scope else 0xB0B
                plan, _ = self.dist.rsample_plan(pr_state, seed=site, idx=db.get("plan_idx"))
                per.append((self.modality_scope, db, emb, latent_goal, seq_feat, plan, self.compute_kl_loss(pp_state, pr_state)))
            # the action decoder sees all modalities at once (shared weights, independent sequences); it returns one loss per
            # modality, each the mean over that modality's own tokens as in the reference (hulc2.py:239-241)
            act_losses = self.action_decoder.loss_segments([p[5] for p in per], [p[2] for p in per], [p[3] for p in per],
                                                           [p[1]["actions"] for p in per], [p[1]["state_info"]["robot_obs"] for p in per])
        # the scalar tail (hulc2.py:400-430) is one launch per direction: total = (sum act + sum kl) / n + beta * clip
        for i, (self.modality_scope, db, emb, latent_goal, seq_feat, plan, kl) in enumerate(per):
            if "lang" in self.modality_scope:
                batch_size["aux_lang"] = 1
                if self.use_clip_auxiliary_loss:
                    if not clip_forked:
                        row0 = plan if (kl_stacked is not None and isinstance(plan, int)) else 0    # (stacked rows: see `stacked_head` above)
                        lang_clip_loss = acc(lang_clip_loss, self.clip_auxiliary_loss(seq_feat, latent_goal, db["use_for_aux_lang_loss"], row0=row0))
                    # hulc2.py:391-394: the epoch mean of train/lang_clip_loss is weighted by the number of masked-in rows (1 when there are
                    # none) — counted by the loss kernel, a device value (no torch.any / torch.sum host round trip)
                    batch_size["aux_lang"] = self._aux_lang_rows
            bs = db["actions"].shape[0]
            batch_size[self.modality_scope] = bs
            total_bs += bs
        n = len(batch)
        if clip_forked:
            cur.wait_stream(side)
            lang_clip_loss.record_stream(cur)
        kl_vec = kl_stacked if kl_stacked is not None else torch.stack([p[6].reshape(()) for p in per])
        act_vec = act_losses if act_losses.dim() == 1 else act_losses.reshape(-1)
        clip_term = lang_clip_loss if (self.use_clip_auxiliary_loss and lang_clip_loss is not None) else None
        total_loss, logs = HF.LossCombineFn.apply(kl_vec, act_vec, clip_term, float(self.clip_auxiliary_loss_beta))
        kl_d, act_d = kl_vec.detach(), act_vec.detach()
        logged = []
        for i, (scope, db, *_rest) in enumerate(per):
            bs = db["actions"].shape[0]
            logged.append((f"train/kl_loss_scaled_{scope}", kl_d[i], dict(on_step=False, on_epoch=True, batch_size=bs)))
            logged.append((f"train/action_loss_{scope}", act_d[i], dict(on_step=False, on_epoch=True, batch_size=bs)))
            logged.append((f"train/total_loss_{scope}", logs[3 + i], dict(on_step=False, on_epoch=True, batch_size=bs)))
        if clip_term is not None:
            logged.append(("train/lang_clip_loss", logs[2], dict(on_step=False, on_epoch=True, batch_size=batch_size.get("aux_lang", 1), sync_dist=True)))
        logged.append(("train/kl_loss", logs[0], dict(on_step=False, on_epoch=True, batch_size=total_bs)))
        logged.append(("train/action_loss", logs[1], dict(on_step=False, on_epoch=True, batch_size=total_bs)))
        logged.append(("train/total_loss", total_loss, dict(on_step=False, on_epoch=True, batch_size=total_bs)))
        return total_loss, logged

    @staticmethod
    def _batchable(mods) -> bool:
        """all modalities carry the same cameras and tensor shapes (the CALVIN vis/lang batches do)"""
        import os
        if len(mods) < 2 or os.environ.get("HULC_NO_MODALITY_BATCHING"):
            return False
        ref = mods[0][1]
        for _, db in mods[1:]:
            if db["actions"].shape != ref["actions"].shape or set(db["rgb_obs"]) != set(ref["rgb_obs"]):
                return False
            if any(db["rgb_obs"][k].shape != ref["rgb_obs"][k].shape for k in ref["rgb_obs"]):
                return False
        return True

    def compute_kl_loss(self, pp_state: State, pr_state: State) -> torch.Tensor:
        """hulc2.py:444-466."""
        return self.dist.kl_balanced(pp_state, pr_state, self.kl_beta, self.kl_balancing_mix)

    def set_kl_beta(self, kl_beta):
        self.kl_beta = kl_beta

    def clip_auxiliary_loss(self, seq_vis_feat, encoded_lang, use_for_aux_loss, row0: int = 0):
        """hulc2.py:472-508; rows with use_for_aux_loss == False are excluded inside the kernel.  row0 > 0: the features are the stacked rows of
        several modalities, the flags describe rows row0 .. and the rows below never take part."""
        if use_for_aux_loss is None:
            use_for_aux_loss = torch.ones(seq_vis_feat.shape[0] - row0, dtype=torch.bool, device=seq_vis_feat.device)
        with kn.site_scope("head"):              # exact-fp32 projections inside a bf16 step (selective precision)
            im, tx = self.proj_vis_lang(seq_vis_feat, encoded_lang)
            loss, self._aux_lang_rows = HF.ClipLossFn.apply(im, tx, use_for_aux_loss, self.logit_scale, int(row0))
            return loss

    # ---- validation and rollout inference on the same kernels ------------------------
    _plan_calls = 0

    def _sample_plan(self, state: State, idx: Optional[torch.Tensor] = None) -> torch.Tensor:
        """dist.sample_latent_plan(dist.get_dist(state)) (distributions.py:23-35, hulc2.py:287,302): a one-hot draw per category,
        no gradient.  A fresh counter-RNG stream per call; `idx` injects the class indices (parity tests)."""
        Hulc2._plan_calls += 1
        with torch.no_grad():
            plan, _ = self.dist.rsample_plan(state, seed=0xC0FFEE00 + Hulc2._plan_calls, idx=idx)
        return plan

    @torch.no_grad()
    def lmp_val(self, perceptual_emb, latent_goal, actions, robot_obs, plan_idx_pp=None, plan_idx_pr=None):
        """hulc2.py:247-334: plans sampled from the prior and the posterior, decoder loss + one sampled action sequence for each,
        KL, per-dimension mean absolute errors and gripper success rates."""
        def metrics(sample_act):
            mae = torch.mean(torch.abs(sample_act[..., :-1] - actions[..., :-1]), 1)             # (batch, 6)
            grip = torch.where(sample_act[..., -1] > 0, 1.0, -1.0)
            return mae, torch.mean((actions[..., -1] == grip).float())

        pp_state = self.plan_proposal(perceptual_emb[:, 0], latent_goal)
        sampled_plan_pp = self._sample_plan(pp_state, plan_idx_pp)
        action_loss_pp, sample_act_pp = self.action_decoder.loss_and_act(sampled_plan_pp, perceptual_emb, latent_goal, actions, robot_obs)
        mae_pp, gripper_sr_pp = metrics(sample_act_pp)
        pr_state, seq_feat = self.plan_recognition(perceptual_emb)
        sampled_plan_pr = self._sample_plan(pr_state, plan_idx_pr)
        action_loss_pr, sample_act_pr = self.action_decoder.loss_and_act(sampled_plan_pr, perceptual_emb, latent_goal, actions, robot_obs)
        mae_pr, gripper_sr_pr = metrics(sample_act_pr)
        kl_loss = self.compute_kl_loss(pp_state, pr_state)
        return (sampled_plan_pp, action_loss_pp, sampled_plan_pr, action_loss_pr, kl_loss, mae_pp, mae_pr, gripper_sr_pp, gripper_sr_pr,
                seq_feat)

    @_kernel_precision
    @torch.no_grad()
    def validation_step(self, batch: Dict[str, Dict], batch_idx: int) -> Dict[str, torch.Tensor]:
        """hulc2.py:510-598: same logged names, returns the sampled plans and episode indices per modality."""
        self._fresh_weight_copies()
        output = {}
        val_total_act_loss_pp = None
        for self.modality_scope, db in batch.items():
            emb = self.perceptual_encoder(db["rgb_obs"], db["depth_obs"], db["robot_obs"])
            latent_goal = self.language_goal(db["lang"]) if "lang" in self.modality_scope else self.visual_goal(emb[:, -1])
            (plan_pp, act_loss_pp, plan_pr, act_loss_pr, kl_loss, mae_pp, mae_pr, grip_pp, grip_pr, seq_feat) = self.lmp_val(
                emb, latent_goal, db["actions"], db["state_info"]["robot_obs"], db.get("plan_idx_pp"), db.get("plan_idx_pr"))
            if "lang" in self.modality_scope and self.use_clip_auxiliary_loss:
                self.log("val/val_pred_clip_loss", self.clip_auxiliary_loss(seq_feat, latent_goal, db["use_for_aux_lang_loss"]), sync_dist=True)
            val_total_act_loss_pp = act_loss_pp if val_total_act_loss_pp is None else val_total_act_loss_pp + act_loss_pp
            m = self.modality_scope
            self.log(f"val_total_mae/{m}_total_mae_pr", mae_pr.mean(), sync_dist=True)
            self.log(f"val_total_mae/{m}_total_mae_pp", mae_pp.mean(), sync_dist=True)
            self.log(f"val_pos_mae/{m}_pos_mae_pr", mae_pr[..., :3].mean(), sync_dist=True)
            self.log(f"val_pos_mae/{m}_pos_mae_pp", mae_pp[..., :3].mean(), sync_dist=True)
            self.log(f"val_orn_mae/{m}_orn_mae_pr", mae_pr[..., 3:6].mean(), sync_dist=True)
            self.log(f"val_orn_mae/{m}_orn_mae_pp", mae_pp[..., 3:6].mean(), sync_dist=True)
            self.log(f"val_kl/{m}_kl_loss", kl_loss, sync_dist=True)
            self.log(f"val_act/{m}_act_loss_pp", act_loss_pp, sync_dist=True)
            self.log(f"val_act/{m}_act_loss_pr", act_loss_pr, sync_dist=True)
            self.log(f"val_grip/{m}_grip_sr_pr", grip_pr, sync_dist=True)
            self.log(f"val_grip/{m}_grip_sr_pp", grip_pp, sync_dist=True)
            n_mod = len(getattr(getattr(getattr(self, "trainer", None), "datamodule", None), "modalities", None) or batch)
            self.log("val_act/action_loss_pp", val_total_act_loss_pp / n_mod, sync_dist=True)
            output[f"sampled_plan_pp_{m}"] = plan_pp
            output[f"sampled_plan_pr_{m}"] = plan_pr
            output[f"idx_{m}"] = db["idx"]
        return output

    def reset(self):
        """hulc2.py:600-606: call at the beginning of a rollout."""
        self.plan = None
        self.latent_goal = None
        self.rollout_step_counter = 0

    @_kernel_precision
    def step(self, obs, goal):
        """hulc2.py:608-628: one control step; a new plan is sampled from the prior every `replan_freq` steps."""
        if self.rollout_step_counter % self.replan_freq == 0:
            if "lang" in goal:
                self.plan, self.latent_goal = self.get_pp_plan_lang(obs, goal)
            else:
                self.plan, self.latent_goal = self.get_pp_plan_vision(obs, goal)
        action = self.predict_with_plan(obs, self.latent_goal, self.plan)
        self.rollout_step_counter += 1
        return action

    @torch.no_grad()
    def predict_with_plan(self, obs, latent_goal, sampled_plan):
        """hulc2.py:630-652."""
        self._fresh_weight_copies()
        emb = self.perceptual_encoder(obs["rgb_obs"], obs["depth_obs"], obs["robot_obs"])
        return self.action_decoder.act(sampled_plan, emb, latent_goal, obs["robot_obs_raw"])

    @torch.no_grad()
    def get_pp_plan_vision(self, obs: dict, goal: dict):
        """hulc2.py:654-683: current and goal frames as a 2-step sequence through the encoders, plan from the prior."""
        assert len(obs["rgb_obs"]) == len(goal["rgb_obs"])
        self._fresh_weight_copies()
        imgs = {k: torch.cat([v, goal["rgb_obs"][k]], dim=1) for k, v in obs["rgb_obs"].items()}     # (1, 2, C, H, W)
        state = torch.cat([obs["robot_obs"], goal["robot_obs"]], dim=1) if "robot_obs" in obs and "robot_obs" in goal else None
        emb = self.perceptual_encoder(imgs, {}, state)
        latent_goal = self.visual_goal(emb[:, -1])
        sampled_plan = self._sample_plan(self.plan_proposal(emb[:, 0], latent_goal))
        self.action_decoder.clear_hidden_state()
        return sampled_plan, latent_goal

    @torch.no_grad()
    def get_pp_plan_lang(self, obs: dict, goal: dict):
        """hulc2.py:685-707."""
        self._fresh_weight_copies()
        emb = self.perceptual_encoder(obs["rgb_obs"], obs["depth_obs"], obs["robot_obs"])
        latent_goal = self.language_goal(goal["lang"])
        sampled_plan = self._sample_plan(self.plan_proposal(emb[:, 0], latent_goal))
        self.action_decoder.clear_hidden_state()
        return sampled_plan, latent_goal

    def on_train_epoch_start(self) -> None:
        logger.info("Start training epoch %s", getattr(self, "current_epoch", "?"))

    def on_train_epoch_end(self, unused=None) -> None:
        logger.info("Finished training epoch %s", getattr(self, "current_epoch", "?"))

    def on_validation_epoch_end(self) -> None:
        logger.info("Finished validation epoch %s", getattr(self, "current_epoch", "?"))
