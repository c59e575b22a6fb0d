"""Training-path shell of the reference's ARModel (neural_lam/models/ar_model.py:
__init__ :30-151, configure_optimizers :191-195, expand_to_batch :204-209,
unroll_prediction :220-267, common_step :269-285, training_step :287-309).
Validation / test / plotting / W&B stay with the reference (out of scope).

It is a pytorch_lightning.LightningModule when Lightning is installed and a
plain nn.Module otherwise; the method names and batch layout are Lightning's.
"""
import os
import torch
import torch.utils.checkpoint
from torch import nn

from .. import glue, metrics

try:  # pragma: no cover - Lightning is not installed in the build image
    import pytorch_lightning as pl

    _Base = pl.LightningModule
except ImportError:
    _Base = nn.Module


def _state_feature_weights(config, datastore):
    """loss_weighting.py:52-106: manual weights if the config carries them,
    uniform 1/n otherwise."""
    n = datastore.get_num_data_vars(category="state")
    weighting = getattr(getattr(config, "training", None), "state_feature_weighting", None)
    weights = getattr(weighting, "weights", None)
    if weights:
        names = datastore.get_vars_names(category="state")
        if set(names) != set(weights):
            raise ValueError("State feature weights must be provided for each state feature")
        return [weights[k] for k in names]
    return [1.0 / n] * n


class _RecomputedPredictStep(torch.autograd.Function):
    """One AR step whose activations are NOT kept: forward runs predict_step under no_grad
    and saves only its three inputs; backward re-runs it with grad enabled and backpropagates
    through the fresh graph (parameter gradients accumulate into .grad as usual).  The fused
    operators keep their activations on ctx (raw device buffers the C ABI reads), which
    torch.utils.checkpoint's saved-tensor hooks cannot see -- hence this explicit form.
    `anchor` is a dummy that requires grad so that the first AR step (whose inputs do not)
    still gets a backward."""

    @staticmethod
    def forward(ctx, anchor, model, prev_state, prev_prev_state, forcing):
        ctx.model = model
        ctx.save_for_backward(prev_state, prev_prev_state, forcing)
        ctx.need = (prev_state.requires_grad, prev_prev_state.requires_grad)
        with torch.no_grad():
            out = model.predict_step(prev_state, prev_prev_state, forcing)[0]
        return out

    @staticmethod
    def backward(ctx, g):
        a, b, f = ctx.saved_tensors
        a = a.detach().requires_grad_(ctx.need[0])
        b = b.detach().requires_grad_(ctx.need[1])
        with torch.enable_grad():
            out = ctx.model.predict_step(a, b, f.detach())[0]
        torch.autograd.backward(out, g)
        return None, None, a.grad, b.grad, None


class ARModel(_Base):
    def __init__(self, args, config, datastore):
        super().__init__()
        if hasattr(self, "save_hyperparameters") and _Base is not nn.Module:
            self.save_hyperparameters(ignore=["datastore"])
        self.args = args
        self._datastore = datastore
        num_state_vars = datastore.get_num_data_vars(category="state")
        num_forcing_vars = datastore.get_num_data_vars(category="forcing")
        da_static = datastore.get_dataarray(category="static", split=None)
        da_stats = datastore.get_standardization_dataarray(category="state")

        arr_static = da_static.transpose("grid_index", "static_feature").values
        self.register_buffer(
            "grid_static_features", torch.tensor(arr_static, dtype=torch.float32), persistent=False
        )
        for key, src in (("state_mean", "state_mean"), ("state_std", "state_std"),
                         ("diff_mean", "state_diff_mean"), ("diff_std", "state_diff_std")):
            self.register_buffer(
                key, torch.tensor(getattr(da_stats, src).values, dtype=torch.float32),
                persistent=False,
            )
        self.feature_weights = torch.tensor(
            _state_feature_weights(config, datastore), dtype=torch.float32
        )
        self.output_std = bool(args.output_std)
        self.ar_checkpoint = bool(getattr(args, "ar_checkpoint", False))
        if self.output_std:
            self.grid_output_dim = 2 * num_state_vars
        else:
            self.grid_output_dim = num_state_vars
            self.register_buffer(
                "per_var_std", self.diff_std / torch.sqrt(self.feature_weights), persistent=False
            )
        self.num_grid_nodes, grid_static_dim = self.grid_static_features.shape
        # ar_model.py:111-116 (kept as in the reference, including 2*grid_output_dim)
        self.grid_dim = (
            2 * self.grid_output_dim
            + grid_static_dim
            + num_forcing_vars
            * (args.num_past_forcing_steps + args.num_future_forcing_steps + 1)
        )
        self.loss = metrics.get_metric(args.loss)
        boundary_mask = torch.tensor(datastore.boundary_mask.values, dtype=torch.float32).unsqueeze(1)
        self.register_buffer("boundary_mask", boundary_mask, persistent=False)
        self.register_buffer("interior_mask", 1.0 - self.boundary_mask, persistent=False)
        self.restore_opt = getattr(args, "restore_opt", False)
        # evaluation state (ar_model.py:133-151); plotting / W&B are out of scope
        self.val_metrics = {"mse": []}
        self.test_metrics = {"mse": [], "mae": []}
        if self.output_std:
            self.test_metrics["output_std"] = []
        self.spatial_loss_maps = []
        self.eval_results = {}   # prefix -> {metric: (pred_steps, d_f) tensor}, rank 0

    def configure_optimizers(self):
        return torch.optim.AdamW(self.parameters(), lr=self.args.lr, betas=(0.9, 0.95))

    @property
    def interior_mask_bool(self):
        return self.interior_mask[:, 0].to(torch.bool)

    @staticmethod
    def expand_to_batch(x, batch_size):
        # same stride-0 view as the reference; the un-expanded (1, N, d) tensor rides along
        # so the fused operators take it directly and its gradient skips autograd's
        # zero-fill + slice-copy + batch-sum of the expand (fused._base)
        base = x.unsqueeze(0)
        out = base.expand(batch_size, -1, -1)
        out._nlam_base = base
        return out

    def predict_step(self, prev_state, prev_prev_state, forcing):
        raise NotImplementedError("No prediction step implemented")

    def unroll_prediction(self, init_states, forcing_features, true_states):
        """init_states (B,2,N,d_f), forcing (B,T,N,d_forcing), true_states (B,T,N,d_f)."""
        prev_prev_state = init_states[:, 0]
        prev_state = init_states[:, 1]
        # args.ar_checkpoint (not a reference option): keep only the states
        # between AR steps and recompute each predict_step in backward
        # (_RecomputedPredictStep), which lifts the ar_steps memory ceiling of BPTT at
        # about one extra forward per step
        ckpt = (getattr(self, "ar_checkpoint", False) and torch.is_grad_enabled()
                and forcing_features.shape[1] > 1 and not self.output_std)
        # the static-feature embeddings do not depend on the state: one evaluation serves every
        # AR step of this rollout (the reference recomputes them per step, base_graph_model.py:
        # 127-130; same values, and their gradient is the sum over the steps either way).  Not
        # with ar_checkpoint, whose recomputed segments must be self-contained.
        share = (not ckpt and forcing_features.shape[1] > 1 and prev_state.is_cuda
                 and hasattr(self, "static_embedders"))
        if share:
            from .. import fused

            self._static_emb_rollout = fused.embed_many(self.static_embedders())
        try:
            return self._unroll(prev_state, prev_prev_state, forcing_features, true_states, ckpt)
        finally:
            self._static_emb_rollout = None

    def _unroll(self, prev_state, prev_prev_state, forcing_features, true_states, ckpt):
        prediction_list, pred_std_list = [], []
        # residual + boundary overwrite in one kernel (glue.StateStep) where predict_step offers it
        import inspect
        fuse_mix = (prev_state.is_cuda and not self.output_std and not ckpt
                    and "boundary_truth" in inspect.signature(self.predict_step).parameters)
        for i in range(forcing_features.shape[1]):
            if ckpt:
                if not hasattr(self, "_ckpt_anchor") or self._ckpt_anchor.device != prev_state.device:
                    self._ckpt_anchor = torch.zeros((), device=prev_state.device, requires_grad=True)
                pred_state = _RecomputedPredictStep.apply(
                    self._ckpt_anchor, self, prev_state, prev_prev_state, forcing_features[:, i])
                pred_std = None
            elif fuse_mix:
                new_state, pred_std = self.predict_step(
                    prev_state, prev_prev_state, forcing_features[:, i],
                    boundary_truth=true_states[:, i])
            else:
                pred_state, pred_std = self.predict_step(
                    prev_state, prev_prev_state, forcing_features[:, i]
                )
            if not fuse_mix:
                new_state = glue.BoundaryMix.apply(pred_state, true_states[:, i], self.boundary_mask)
            prediction_list.append(new_state)
            if self.output_std:
                pred_std_list.append(pred_std)
            prev_prev_state = prev_state
            prev_state = new_state
        prediction = (prediction_list[0].unsqueeze(1) if len(prediction_list) == 1
                      else torch.stack(prediction_list, dim=1))
        if self.output_std:
            pred_std = torch.stack(pred_std_list, dim=1)
        else:
            pred_std = self.per_var_std
        return prediction, pred_std

    def common_step(self, batch):
        init_states, target_states, forcing_features, batch_times = batch
        prediction, pred_std = self.unroll_prediction(init_states, forcing_features, target_states)
        return prediction, target_states, pred_std, batch_times

    def _masked_loss_consts(self, device):
        if not hasattr(self, "_loss_consts") or self._loss_consts[0].device != device:
            keep = self.interior_mask[:, 0].contiguous()
            w = (
                1.0 / self.per_var_std**2 if self.loss is metrics.wmse
                else torch.ones_like(self.per_var_std)
            ).contiguous()
            self._loss_consts = (keep, w, float(keep.sum().item()))
        return self._loss_consts

    def training_step(self, batch):
        init_states, forcing = batch[0], batch[2]
        wmse_like = not self.output_std and self.loss in (metrics.wmse, metrics.mse)
        # the loss target of AR step t is the boundary truth of the same step: on the device path
        # each predict_step hands back its loss term from the state-step kernel itself
        # (glue.StateStepLoss; NLAM_FUSE_LOSS=0: the separate loss kernel over the stacked prediction)
        tap = (wmse_like and init_states.is_cuda and not getattr(self, "ar_checkpoint", False)
               and os.environ.get("NLAM_FUSE_LOSS", "1") != "0")
        if tap:
            keep, w, n_keep = self._masked_loss_consts(init_states.device)
            self._loss_tap = (keep, w, 1.0 / (n_keep * init_states.shape[0] * forcing.shape[1]))
            self._loss_terms = []
        try:
            prediction, target, pred_std, _ = self.common_step(batch)
            terms = self._loss_terms if tap else []
        finally:
            self._loss_tap, self._loss_terms = None, []
        if terms:
            assert len(terms) == forcing.shape[1]
            batch_loss = terms[0]
            for t in terms[1:]:
                batch_loss = batch_loss + t
        elif wmse_like:
            # fused masked loss kernel (no boolean-mask gather, no host sync)
            keep, w, n_keep = self._masked_loss_consts(prediction.device)
            lead = prediction.numel() // (prediction.shape[-1] * prediction.shape[-2])
            batch_loss = glue.MaskedWMSE.apply(prediction, target, keep, w, 1.0 / (n_keep * lead))
        elif self.output_std and self.loss is metrics.nll and prediction.is_cuda:
            if not hasattr(self, "_nll_consts") or self._nll_consts[0].device != prediction.device:
                keep = self.interior_mask[:, 0].contiguous()
                self._nll_consts = (keep, float(keep.sum().item()))
            keep, n_keep = self._nll_consts
            lead = prediction.numel() // (prediction.shape[-1] * prediction.shape[-2])
            batch_loss = glue.MaskedNLL.apply(prediction, target, pred_std, keep,
                                              1.0 / (n_keep * lead))
        else:
            batch_loss = torch.mean(
                self.loss(prediction, target, pred_std, mask=self.interior_mask_bool)
            )
        if hasattr(self, "log_dict") and _Base is not nn.Module:
            self.log_dict({"train_loss": batch_loss}, prog_bar=True, on_step=True, on_epoch=True,
                          sync_dist=True, batch_size=batch[0].shape[0])
        return batch_loss


    # ------------------------------------------------------------ evaluation
    # Reference ar_model.py:311-452, 610-696 without the plotting / W&B parts: same
    # metrics, same logged keys, same aggregation (mean over samples, sqrt for *mse ->
    # *rmse, rescale by state_std); results are kept in self.eval_results.
    def _log(self, values, batch_size=None):
        if hasattr(self, "log_dict") and getattr(self, "_trainer", None) is not None:
            self.log_dict(values, on_step=False, on_epoch=True, sync_dist=True, batch_size=batch_size)
        self.last_logged = {k: (float(v) if torch.is_tensor(v) and v.numel() == 1 else v)
                            for k, v in values.items()}

    def all_gather_cat(self, tensor_to_gather):
        """(d1, ...) on K ranks -> (K d1, ...) (ar_model.py:311-320)."""
        if hasattr(self, "all_gather") and getattr(self, "_trainer", None) is not None:
            return self.all_gather(tensor_to_gather).flatten(0, 1)
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            world = torch.distributed.get_world_size()
            if world > 1:
                parts = [torch.empty_like(tensor_to_gather) for _ in range(world)]
                torch.distributed.all_gather(parts, tensor_to_gather.contiguous())
                return torch.cat(parts, dim=0)
        return tensor_to_gather

    def _is_rank_zero(self):
        tr = getattr(self, "_trainer", None)
        if tr is not None:
            return bool(tr.is_global_zero)
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            return torch.distributed.get_rank() == 0
        return True

    def _steps_to_log(self):
        return list(getattr(self.args, "val_steps_to_log", [1]))

    def _eval_losses(self, batch, prefix, clip_steps):
        prediction, target, pred_std, _ = self.common_step(batch)
        time_step_loss = torch.mean(
            self.loss(prediction, target, pred_std, mask=self.interior_mask_bool), dim=0)
        steps = [st for st in self._steps_to_log() if not clip_steps or st <= len(time_step_loss)]
        log = {f"{prefix}_loss_unroll{st}": time_step_loss[st - 1] for st in steps}
        log[f"{prefix}_mean_loss"] = torch.mean(time_step_loss)
        self._log(log, batch_size=batch[0].shape[0])
        return prediction, target, pred_std

    @torch.no_grad()
    def validation_step(self, batch, batch_idx=0):
        """ar_model.py:324-361."""
        prediction, target, pred_std = self._eval_losses(batch, "val", clip_steps=True)
        self.val_metrics["mse"].append(
            metrics.mse(prediction, target, pred_std, mask=self.interior_mask_bool, sum_vars=False))

    def on_validation_epoch_end(self):
        self.aggregate_metrics(self.val_metrics, prefix="val")
        for lst in self.val_metrics.values():
            lst.clear()

    @torch.no_grad()
    def test_step(self, batch, batch_idx=0):
        """ar_model.py:375-452 (metrics and spatial loss maps; no example plots)."""
        prediction, target, pred_std = self._eval_losses(batch, "test", clip_steps=False)
        for name in ("mse", "mae"):
            self.test_metrics[name].append(metrics.get_metric(name)(
                prediction, target, pred_std, mask=self.interior_mask_bool, sum_vars=False))
        if self.output_std:
            self.test_metrics["output_std"].append(
                torch.mean(pred_std[..., self.interior_mask_bool, :], dim=-2))
        spatial = self.loss(prediction, target, pred_std, average_grid=False)
        self.spatial_loss_maps.append(spatial[:, [st - 1 for st in self._steps_to_log()]])

    def aggregate_metrics(self, metrics_dict, prefix):
        """ar_model.py:610-644 without the figures: gathered over ranks, averaged over the
        evaluated samples, *mse -> *rmse, rescaled to physical units by state_std."""
        out = {}
        for name, vals in metrics_dict.items():
            if not vals:
                continue
            gathered = self.all_gather_cat(torch.cat(vals, dim=0))     # (N_eval, steps, d_f)
            if self._is_rank_zero():
                avg = torch.mean(gathered, dim=0)
                if "mse" in name:
                    avg, name = torch.sqrt(avg), name.replace("mse", "rmse")
                out[f"{prefix}_{name}"] = avg * self.state_std
        if self._is_rank_zero():
            self.eval_results[prefix] = out
        return out

    def on_test_epoch_end(self):
        """ar_model.py:646-696: metrics + mean spatial loss maps (kept as tensors)."""
        out = self.aggregate_metrics(self.test_metrics, prefix="test")
        if self.spatial_loss_maps:
            maps = self.all_gather_cat(torch.cat(self.spatial_loss_maps, dim=0))
            if self._is_rank_zero():
                out["test_mean_spatial_loss"] = torch.mean(maps, dim=0)   # (N_log, N_grid)
        for lst in self.test_metrics.values():
            lst.clear()
        self.spatial_loss_maps.clear()
        return out

    def on_load_checkpoint(self, checkpoint):
        """ar_model.py:698-720: checkpoints from before the encoder refactoring keep the grid
        MLP under g2m_gnn.grid_mlp.*; optionally drop the optimizer state."""
        sd = checkpoint["state_dict"]
        for old in [k for k in sd if k.startswith("g2m_gnn.grid_mlp")]:
            sd[old.replace("g2m_gnn.grid_mlp", "encoding_grid_mlp")] = sd.pop(old)
        if not self.restore_opt and "optimizer_states" in checkpoint:
            opt = self.configure_optimizers()
            checkpoint["optimizer_states"] = [opt.state_dict()]
