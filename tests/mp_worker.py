"""Worker for the multi-rank CPU test: runs the oracle on one rank's element block
with torch.distributed (gloo) providing the halo sums and all-reduces, i.e. the
same decomposition the HIP path uses with RCCL.  Launched by
tests/test_multirank_cpu.py through torch.distributed.run."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class GlooComm:
    def __init__(self, prob):
        self.rank, self.n = dist.get_rank(), dist.get_world_size()
        self.prob = prob
        self.nbr_rank, self.nbr_nodes = prob.neighbors()

    def allreduce_sum(self, v):
        t = torch.tensor([float(v)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t.item())

    def allreduce_min(self, v):
        t = torch.tensor([float(v)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return float(t.item())

    def halo_sum(self, arr, ncomp):
        """in-place sum of shared nodes; contributions added in ascending rank order"""
        N = self.prob.N
        v = arr.reshape(ncomp, N)
        send = [torch.from_numpy(np.ascontiguousarray(v[:, nodes])) for nodes in self.nbr_nodes]
        recv = [torch.empty_like(s) for s in send]
        reqs = []
        for k, r in enumerate(self.nbr_rank):
            reqs.append(dist.isend(send[k], r))
            reqs.append(dist.irecv(recv[k], r))
        for q in reqs:
            q.wait()
        contrib = {}
        for k, r in enumerate(self.nbr_rank):
            rk = recv[k].numpy()
            for j, node in enumerate(self.nbr_nodes[k]):
                contrib.setdefault(int(node), []).append((r, rk[:, j]))
        for node, lst in contrib.items():
            lst.append((self.rank, v[:, node].copy()))
            lst.sort(key=lambda t: t[0])
            s = lst[0][1].copy()
            for _, c in lst[1:]:
                s = s + c
            v[:, node] = s


def main():
    dist.init_process_group("gloo")
    from oracle.driver import run
    from oracle.fem import Problem
    cfg = json.loads(sys.argv[1])
    out_path = sys.argv[2]
    rank, world = dist.get_rank(), dist.get_world_size()
    prob = Problem(mesh=cfg["mesh"], rs=cfg["rs"], order_v=cfg["ok"], order_e=cfg["ot"], problem=cfg["problem"],
                   rank=rank, pgrid=cfg["pgrid"])
    comm = GlooComm(prob)
    r = run(prob, t_final=cfg["tf"], cg_tol=cfg["cgt"], max_steps=cfg["ms"], probe_steps=tuple(cfg["probes"]),
            comm=comm)
    if rank == 0:
        json.dump(dict(probes={str(k): v for k, v in r["probes"].items()}, steps=r["steps"],
                       repeats=r["repeats"], last=r["last"]["e_norm"] if r["last"] else None), open(out_path, "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
