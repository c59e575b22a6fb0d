"""Hierarchical base (reference models/base_hi_graph_model.py:12-235): per-level
embedders, mesh-init up-sweep, hi_processor_step hook, read-out down-sweep."""
from torch import nn

from .. import glue, utils
from ..interaction_net import InteractionNet
from .base_graph_model import BaseGraphModel


class BaseHiGraphModel(BaseGraphModel):
    def __init__(self, args, config, datastore):
        super().__init__(args, config=config, datastore=datastore)
        self.num_levels = len(self.mesh_static_features)
        self.level_mesh_sizes = [f.shape[0] for f in self.mesh_static_features]
        mesh_dim = self.mesh_static_features[0].shape[1]
        mesh_same_dim = self.m2m_features[0].shape[1]
        mesh_up_dim = self.mesh_up_features[0].shape[1]
        mesh_down_dim = self.mesh_down_features[0].shape[1]
        L = self.num_levels
        end = self.mlp_blueprint_end
        self.mesh_embedders = nn.ModuleList([utils.make_mlp([mesh_dim] + end) for _ in range(L)])
        self.mesh_same_embedders = nn.ModuleList(
            [utils.make_mlp([mesh_same_dim] + end) for _ in range(L)]
        )
        self.mesh_up_embedders = nn.ModuleList(
            [utils.make_mlp([mesh_up_dim] + end) for _ in range(L - 1)]
        )
        self.mesh_down_embedders = nn.ModuleList(
            [utils.make_mlp([mesh_down_dim] + end) for _ in range(L - 1)]
        )
        self.mesh_init_gnns = nn.ModuleList(
            [
                InteractionNet(ei, args.hidden_dim, hidden_layers=args.hidden_layers)
                for ei in self.mesh_up_edge_index
            ]
        )
        self.mesh_read_gnns = nn.ModuleList(
            [
                InteractionNet(ei, args.hidden_dim, hidden_layers=args.hidden_layers,
                               update_edges=False)
                for ei in self.mesh_down_edge_index
            ]
        )

    def get_num_mesh(self):
        num_mesh_nodes = sum(f.shape[0] for f in self.mesh_static_features)
        return num_mesh_nodes, num_mesh_nodes - self.mesh_static_features[0].shape[0]

    def static_embedders(self):
        items = super().static_embedders()
        for l, (emb, f) in enumerate(zip(self.mesh_embedders, self.mesh_static_features)):
            items.append((f"mesh{l}", emb, f))
        for l, (emb, f) in enumerate(zip(self.mesh_same_embedders, self.m2m_features)):
            items.append((f"same{l}", emb, f))
        for l, (emb, f) in enumerate(zip(self.mesh_up_embedders, self.mesh_up_features)):
            items.append((f"up{l}", emb, f))
        for l, (emb, f) in enumerate(zip(self.mesh_down_embedders, self.mesh_down_features)):
            items.append((f"down{l}", emb, f))
        return items

    def embedd_mesh_nodes(self):
        return self.static_emb("mesh0", self.mesh_embedders[0], self.mesh_static_features[0])

    def process_step(self, mesh_rep):
        B = mesh_rep.shape[0]
        L = self.num_levels
        mesh_rep_levels = [mesh_rep] + [
            self.expand_to_batch(
                self.static_emb(f"mesh{l}", self.mesh_embedders[l], self.mesh_static_features[l]), B)
            for l in range(1, L)
        ]
        mesh_same_rep = [
            self.expand_to_batch(self.static_emb(f"same{l}", emb, f), B)
            for l, (emb, f) in enumerate(zip(self.mesh_same_embedders, self.m2m_features))
        ]
        mesh_up_rep = [
            self.expand_to_batch(self.static_emb(f"up{l}", emb, f), B)
            for l, (emb, f) in enumerate(zip(self.mesh_up_embedders, self.mesh_up_features))
        ]
        mesh_down_rep = [
            self.expand_to_batch(self.static_emb(f"down{l}", emb, f), B)
            for l, (emb, f) in enumerate(zip(self.mesh_down_embedders, self.mesh_down_features))
        ]
        # mesh init: level l-1 -> l for l = 1..L-1
        for level_l, gnn in enumerate(self.mesh_init_gnns, start=1):
            # (level l-1 is the sender here and an input of the processor later: glue.tee joins
            # the two gradients inside this net's backward instead of a torch add)
            snd, mesh_rep_levels[level_l - 1] = glue.tee(mesh_rep_levels[level_l - 1])
            mesh_rep_levels[level_l], mesh_up_rep[level_l - 1] = gnn(
                snd, mesh_rep_levels[level_l], mesh_up_rep[level_l - 1]
            )
        mesh_rep_levels, _, _, mesh_down_rep = self.hi_processor_step(
            mesh_rep_levels, mesh_same_rep, mesh_up_rep, mesh_down_rep
        )
        # read out: level l+1 -> l for l = L-2..0
        for level_l, gnn in zip(range(L - 2, -1, -1), reversed(self.mesh_read_gnns)):
            mesh_rep_levels[level_l] = gnn(
                mesh_rep_levels[level_l + 1], mesh_rep_levels[level_l], mesh_down_rep[level_l]
            )
        return mesh_rep_levels[0]

    def hi_processor_step(self, mesh_rep_levels, mesh_same_rep, mesh_up_rep, mesh_down_rep):
        raise NotImplementedError("hi_process_step not implemented")
