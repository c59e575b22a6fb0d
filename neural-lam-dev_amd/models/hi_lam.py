"""Hi-LAM (reference models/hi_lam.py:11-207): per processor layer one
sequential down sweep then one up sweep through the mesh hierarchy."""
from torch import nn

from .. import glue
from ..interaction_net import InteractionNet
from .base_hi_graph_model import BaseHiGraphModel


class HiLAM(BaseHiGraphModel):
    def __init__(self, args, config, datastore):
        super().__init__(args, config=config, datastore=datastore)
        P = args.processor_layers
        # construction order as hi_lam.py:22-35 (fixes the init RNG stream)
        self.mesh_down_gnns = nn.ModuleList([self.make_down_gnns(args) for _ in range(P)])
        self.mesh_down_same_gnns = nn.ModuleList([self.make_same_gnns(args) for _ in range(P)])
        self.mesh_up_gnns = nn.ModuleList([self.make_up_gnns(args) for _ in range(P)])
        self.mesh_up_same_gnns = nn.ModuleList([self.make_same_gnns(args) for _ in range(P)])

    def _gnns(self, args, edge_indices, tag):
        nets = nn.ModuleList(
            [InteractionNet(ei, args.hidden_dim, hidden_layers=args.hidden_layers)
             for ei in edge_indices]
        )
        for level, net in enumerate(nets):   # profiler labels: same0 / up1 / down0 ...
            net.tables.tag = f"{tag}{level}"
        return nets

    def make_same_gnns(self, args):
        return self._gnns(args, self.m2m_edge_index, "same")

    def make_up_gnns(self, args):
        return self._gnns(args, self.mesh_up_edge_index, "up")

    def make_down_gnns(self, args):
        return self._gnns(args, self.mesh_down_edge_index, "down")

    def mesh_down_step(self, nodes, same, down, down_gnns, same_gnns):
        """hi_lam.py:82-124: same(L-1); then for l = L-2..0: down(l+1->l), same(l)."""
        top = self.num_levels - 1
        nodes[top], same[top] = same_gnns[top](nodes[top], nodes[top], same[top])
        for l in range(top - 1, -1, -1):
            # nodes[l + 1] has two consumers: this net's sender side and, later, the receiver
            # side of the up net into that level.  glue.tee joins their gradients inside the
            # sender-side store of this net's backward instead of a torch add per level and sweep
            # (18 per Hi-LAM step)
            snd, nodes[l + 1] = glue.tee(nodes[l + 1])
            new, down[l] = down_gnns[l](snd, nodes[l], down[l])
            nodes[l], same[l] = same_gnns[l](new, new, same[l])
        return nodes, same, down

    def mesh_up_step(self, nodes, same, up, up_gnns, same_gnns):
        """hi_lam.py:126-163: same(0); then for l = 1..L-1: up(l-1->l), same(l)."""
        nodes[0], same[0] = same_gnns[0](nodes[0], nodes[0], same[0])
        for l in range(1, self.num_levels):
            # (nodes[l - 1]: sender here, receiver of the next layer's down net -- see above)
            snd, nodes[l - 1] = glue.tee(nodes[l - 1])
            new, up[l - 1] = up_gnns[l - 1](snd, nodes[l], up[l - 1])
            nodes[l], same[l] = same_gnns[l](new, new, same[l])
        return nodes, same, up

    def hi_processor_step(self, mesh_rep_levels, mesh_same_rep, mesh_up_rep, mesh_down_rep):
        for down_gnns, down_same, up_gnns, up_same in zip(
            self.mesh_down_gnns, self.mesh_down_same_gnns, self.mesh_up_gnns,
            self.mesh_up_same_gnns,
        ):
            mesh_rep_levels, mesh_same_rep, mesh_down_rep = self.mesh_down_step(
                mesh_rep_levels, mesh_same_rep, mesh_down_rep, down_gnns, down_same
            )
            mesh_rep_levels, mesh_same_rep, mesh_up_rep = self.mesh_up_step(
                mesh_rep_levels, mesh_same_rep, mesh_up_rep, up_gnns, up_same
            )
        return mesh_rep_levels, mesh_same_rep, mesh_up_rep, mesh_down_rep
