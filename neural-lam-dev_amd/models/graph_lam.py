"""GraphLAM (reference models/graph_lam.py:12-91): flat / multiscale mesh,
processor = processor_layers x m2m InteractionNet with send = rec = mesh."""
from torch import nn

from .. import fused, utils
from ..interaction_net import InteractionNet
from .base_graph_model import BaseGraphModel


class ProcessorSequential(nn.Module):
    """Chains m2m InteractionNets as `mesh, mesh, edge -> mesh, edge`; children
    are named module_<i>, the key layout torch_geometric.nn.Sequential gives
    the reference's checkpoints (graph_lam.py:51-57)."""

    def __init__(self, nets):
        super().__init__()
        for i, net in enumerate(nets):
            self.add_module(f"module_{i}", net)
        self._n = len(nets)

    def __len__(self):
        return self._n

    def __iter__(self):
        return (getattr(self, f"module_{i}") for i in range(self._n))

    def forward(self, mesh_rep, edge_rep):
        nets = list(self)
        if fused.chain_eligible(nets, mesh_rep, edge_rep):
            return fused.apply_chain(nets, mesh_rep, edge_rep)
        for net in self:
            mesh_rep, edge_rep = net(mesh_rep, mesh_rep, edge_rep)
        return mesh_rep, edge_rep


class GraphLAM(BaseGraphModel):
    def __init__(self, args, config, datastore):
        super().__init__(args, config=config, datastore=datastore)
        assert not self.hierarchical, "GraphLAM does not use a hierarchical mesh graph"
        mesh_dim = self.mesh_static_features.shape[1]
        _, m2m_dim = self.m2m_features.shape
        self.mesh_embedder = utils.make_mlp([mesh_dim] + self.mlp_blueprint_end)
        self.m2m_embedder = utils.make_mlp([m2m_dim] + self.mlp_blueprint_end)
        self.processor = ProcessorSequential(
            [
                InteractionNet(self.m2m_edge_index, args.hidden_dim,
                               hidden_layers=args.hidden_layers, aggr=args.mesh_aggr)
                for _ in range(args.processor_layers)
            ]
        )

        for net in self.processor:
            net.tables.tag = "m2m"
        self.mesh_embedder.tag, self.m2m_embedder.tag = "mesh_embedder", "m2m_embedder"

    def get_num_mesh(self):
        return self.mesh_static_features.shape[0], 0

    def static_embedders(self):
        return super().static_embedders() + [
            ("mesh0", self.mesh_embedder, self.mesh_static_features),
            ("m2m", self.m2m_embedder, self.m2m_features)]

    def embedd_mesh_nodes(self):
        return self.static_emb("mesh0", self.mesh_embedder, self.mesh_static_features)

    def process_step(self, mesh_rep):
        batch_size = mesh_rep.shape[0]
        m2m_emb = self.static_emb("m2m", self.m2m_embedder, self.m2m_features)
        mesh_rep, _ = self.processor(mesh_rep, self.expand_to_batch(m2m_emb, batch_size))
        return mesh_rep
