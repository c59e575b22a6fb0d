"""Encode-process-decode base (reference models/base_graph_model.py:12-177):
same sub-module names, construction order (=> same default-init RNG stream)
and predict_step contract; the GNN / MLP blocks are the HIP modules."""
import torch

from .. import fused, glue, utils, wide
from ..interaction_net import InteractionNet
from .ar_model import ARModel


class BaseGraphModel(ARModel):
    def __init__(self, args, config, datastore):
        super().__init__(args, config=config, datastore=datastore)
        graph_dir_path = datastore.root_path / "graph" / args.graph
        self.hierarchical, graph_ldict = utils.load_graph(graph_dir_path=graph_dir_path)
        for name, attr_value in graph_ldict.items():
            if isinstance(attr_value, torch.Tensor):
                self.register_buffer(name, attr_value, persistent=False)
            else:
                setattr(self, name, attr_value)
        self.num_mesh_nodes, _ = self.get_num_mesh()
        self.g2m_edges, g2m_dim = self.g2m_features.shape
        self.m2g_edges, m2g_dim = self.m2g_features.shape

        self.mlp_blueprint_end = [args.hidden_dim] * (args.hidden_layers + 1)
        self.grid_embedder = utils.make_mlp([self.grid_dim] + self.mlp_blueprint_end)
        self.g2m_embedder = utils.make_mlp([g2m_dim] + self.mlp_blueprint_end)
        self.m2g_embedder = utils.make_mlp([m2g_dim] + self.mlp_blueprint_end)
        self.g2m_gnn = InteractionNet(
            self.g2m_edge_index, args.hidden_dim, hidden_layers=args.hidden_layers,
            update_edges=False,
        )
        self.encoding_grid_mlp = utils.make_mlp([args.hidden_dim] + self.mlp_blueprint_end)
        self.m2g_gnn = InteractionNet(
            self.m2g_edge_index, args.hidden_dim, hidden_layers=args.hidden_layers,
            update_edges=False,
        )
        self.g2m_gnn.tables.tag, self.m2g_gnn.tables.tag = "g2m", "m2g"
        for name in ("grid_embedder", "g2m_embedder", "m2g_embedder", "encoding_grid_mlp"):
            getattr(self, name).tag = name
        self.output_map = utils.make_mlp(
            [args.hidden_dim] * (args.hidden_layers + 1) + [self.grid_output_dim], layer_norm=False
        )
        self.output_map.tag = "output_map"

    def get_num_mesh(self):
        raise NotImplementedError("get_num_mesh not implemented")

    def embedd_mesh_nodes(self):
        raise NotImplementedError("embedd_mesh_nodes not implemented")

    def process_step(self, mesh_rep):
        raise NotImplementedError("process_step not implemented")

    # ---- static-feature embedders ---------------------------------------------------------
    # The reference applies each embedder where its output is first needed (base_graph_model.py:
    # 127-130, graph_lam.py:84-88, base_hi_graph_model.py:137-166).  They are independent small
    # MLPs on batch-invariant static features, so one predict_step evaluates all of them in
    # multi-problem launches up front (fused.embed_many) and the call sites read the results.
    def static_embedders(self):
        """[(key, module, features)] of every static-feature embedder of the model."""
        return [("g2m", self.g2m_embedder, self.g2m_features),
                ("m2g", self.m2g_embedder, self.m2g_features)]

    def static_emb(self, key, module, features):
        cache = getattr(self, "_static_emb", None)
        if cache is not None and key in cache:
            return cache[key]
        return module(features)

    def predict_step(self, prev_state, prev_prev_state, forcing, boundary_truth=None):
        """X_{t-1}, X_t, forcing -> X_{t+1}  (base_graph_model.py:106-177).  With boundary_truth
        (device path without output_std only: the rollout's use), the boundary overwrite of
        ar_model.py:244-247 is applied in the same kernel as the state residual and the returned
        state is the rollout's new state."""
        defer = (prev_state.is_cuda and torch.is_grad_enabled() and self.training
                 and not getattr(self, "ar_checkpoint", False))
        if defer:   # (hidden 128 / 256: one slab-reduction flush per AR step, wide.defer_begin)
            wide.defer_begin(self)
        shared = getattr(self, "_static_emb_rollout", None)   # set by a multi-step rollout
        if shared is not None:
            self._static_emb = shared
        else:
            self._static_emb = fused.embed_many(self.static_embedders()) if prev_state.is_cuda else None
        try:
            return self._predict_step(prev_state, prev_prev_state, forcing, boundary_truth)
        finally:
            self._static_emb = None
            fused.PRE.clear()
            if defer:
                wide.defer_end()

    def _predict_step(self, prev_state, prev_prev_state, forcing, boundary_truth=None):
        batch_size = prev_state.shape[0]
        srcs = (prev_state, prev_prev_state, forcing,
                self.expand_to_batch(self.grid_static_features, batch_size))
        if prev_state.is_cuda:
            # the whole grid-side chain up to the m2g receiver projection in one pass over the grid
            # rows (csrc/fused16_grid.hip); the modules below adopt its outputs (fused.PRE)
            if fused.grid_encode_eligible(self, srcs):
                fused.grid_encode(self, srcs)
            grid_features = glue.ConcatRows.apply(*srcs)   # one kernel, static features in place
        else:
            grid_features = torch.cat(srcs, dim=-1)
        grid_emb = self.grid_embedder(grid_features)
        # two consumers (the g2m senders and the grid's own encoding MLP): their gradients meet in
        # the g2m projection backward instead of a full-size elementwise add (glue.Tee)
        grid_emb_s, grid_emb = glue.tee(grid_emb) if prev_state.is_cuda else (grid_emb, grid_emb)
        g2m_emb = self.static_emb("g2m", self.g2m_embedder, self.g2m_features)
        m2g_emb = self.static_emb("m2g", self.m2g_embedder, self.m2g_features)
        mesh_emb = self.embedd_mesh_nodes()

        mesh_rep = self.g2m_gnn(
            grid_emb_s, self.expand_to_batch(mesh_emb, batch_size),
            self.expand_to_batch(g2m_emb, batch_size),
        )
        # grid_rep = grid_emb + MLP(grid_emb): residual fused into the MLP kernel
        grid_rep = self.encoding_grid_mlp(grid_emb, res=grid_emb)
        mesh_rep = self.process_step(mesh_rep)
        grid_rep = self.m2g_gnn(mesh_rep, grid_rep, self.expand_to_batch(m2g_emb, batch_size))
        net_output = self.output_map(grid_rep)

        assert boundary_truth is None or (prev_state.is_cuda and not self.output_std)
        if self.output_std:
            # chunk + softplus + rescale + residual in one kernel (glue.StdHead)
            return glue.StdHead.apply(prev_state, net_output, self.diff_std, self.diff_mean)
        tap = getattr(self, "_loss_tap", None)
        if boundary_truth is not None and tap is not None:
            # training: this AR step's loss term from the same pass (ARModel.training_step sums them)
            keep, w, lscale = tap
            new_state, term = glue.StateStepLoss.apply(
                prev_state, net_output, boundary_truth, self.boundary_mask, self.diff_std,
                self.diff_mean, keep, w, lscale)
            self._loss_terms.append(term)
            return new_state, None
        if boundary_truth is not None:
            return glue.StateStep.apply(prev_state, net_output, boundary_truth, self.boundary_mask,
                                        self.diff_std, self.diff_mean), None
        new_state = glue.StateResidual.apply(
            prev_state, net_output, self.diff_std, self.diff_mean
        )
        return new_state, None
