"""HiLAMParallel (reference models/hi_lam_parallel.py:12-99): one InteractionNet
over the union of all mesh edges per processor layer, with SplitMLPs per edge
set (same levels, up, down) and per node level."""
import torch

from ..interaction_net import InteractionNet
from .base_hi_graph_model import BaseHiGraphModel
from .graph_lam import ProcessorSequential


class HiLAMParallel(BaseHiGraphModel):
    def __init__(self, args, config, datastore):
        super().__init__(args, config=config, datastore=datastore)
        total_edge_index_list = (
            list(self.m2m_edge_index) + list(self.mesh_up_edge_index)
            + list(self.mesh_down_edge_index)
        )
        total_edge_index = torch.cat(total_edge_index_list, dim=1)
        self.edge_split_sections = [ei.shape[1] for ei in total_edge_index_list]
        if args.processor_layers == 0:
            self.processor = lambda x, edge_attr: (x, edge_attr)
        else:
            self.processor = ProcessorSequential(
                [
                    InteractionNet(total_edge_index, args.hidden_dim,
                                   hidden_layers=args.hidden_layers,
                                   edge_chunk_sizes=self.edge_split_sections,
                                   aggr_chunk_sizes=self.level_mesh_sizes)
                    for _ in range(args.processor_layers)
                ]
            )

    def hi_processor_step(self, mesh_rep_levels, mesh_same_rep, mesh_up_rep, mesh_down_rep):
        L = self.num_levels
        mesh_rep = torch.cat(mesh_rep_levels, dim=1)
        mesh_edge_rep = torch.cat(mesh_same_rep + mesh_up_rep + mesh_down_rep, dim=1)
        mesh_rep, mesh_edge_rep = self.processor(mesh_rep, mesh_edge_rep)
        mesh_rep_levels = list(torch.split(mesh_rep, self.level_mesh_sizes, dim=1))
        sections = torch.split(mesh_edge_rep, self.edge_split_sections, dim=1)
        return (mesh_rep_levels, list(sections[:L]), list(sections[L : 2 * L - 1]),
                list(sections[2 * L - 1 :]))
