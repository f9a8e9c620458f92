"""Worker of tests/test_parallel_gloo.py: one rank of a world_size-N gloo job on the CPU.
Checks the sharded path by construction: partition, halo plan, halo exchange, distributed
assembly/SpMV/dot equal to the serial oracle."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from cfd_hemodynamic_amd.parallel import PartComm  # noqa: E402
from oracle import orc  # noqa: E402
from util import dfg_case, make_oracle  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    os.environ["CFDH_ORACLE_THREADS"] = "1"
    case = dfg_case(8)
    mesh = case.mesh
    nv = mesh.num_vertices
    comm = PartComm(rank, world, "host")
    part = comm.make_part(mesh)

    # 1. halo exchange delivers the owners' values
    gid = np.stack([part.l2g * 1.0, part.l2g * 2.0 + 0.5, -part.l2g * 1.0], axis=1)  # (ux, uy, p) coded by global id
    s = torch.from_numpy(np.ascontiguousarray(gid[part.send_idx]).reshape(-1))
    r = torch.zeros(3 * part.ng, dtype=torch.float64)
    comm.host_exchange(s, r)
    assert np.array_equal(r.numpy().reshape(-1, 3), gid[part.nvo:]), "ghost values differ from the owners'"

    # 2. local assembly on the overlapping part reproduces the owned rows of the serial system
    rng = np.random.default_rng(7)
    xg = 0.1 * rng.standard_normal(3 * nv)
    ung = 0.1 * rng.standard_normal(2 * nv)
    O = make_oracle(case)
    O.set_un(ung)
    Fg = O.assemble(xg)
    Jg = O.csr()
    OL = orc.Oracle(part.x, part.cells, part.facet_cells, part.facet_local, case.dt, case.rho, case.mu, case.f)
    for field, nodes, vals in case.bcs:
        loc = part.g2l[nodes]
        keep = loc >= 0
        (OL.add_bc_u if field == 0 else OL.add_bc_p)(loc[keep].astype(np.int32), vals[keep])
    nl = part.nv
    l2g = part.l2g
    xl = np.concatenate([xg[: 2 * nv].reshape(-1, 2)[l2g].ravel(), xg[2 * nv:][l2g]])
    OL.set_un(ung.reshape(-1, 2)[l2g].ravel())
    Fl = OL.assemble(xl)
    Jl = OL.csr()
    own = np.arange(part.nvo)
    rows_l = np.concatenate([2 * own, 2 * own + 1, 2 * nl + own])
    og = part.owned_global
    rows_g = np.concatenate([2 * og, 2 * og + 1, 2 * nv + og])
    assert np.abs(Fl[rows_l] - Fg[rows_g]).max() <= 1e-13 * np.abs(Fg).max(), "owned residual rows differ"
    # columns: local dof -> global dof
    cmap = np.concatenate([np.stack([2 * l2g, 2 * l2g + 1], 1).ravel(), 2 * nv + l2g])
    A = Jl[rows_l].tocoo()
    import scipy.sparse as sp
    Ag = sp.csr_matrix((A.data, (A.row, cmap[A.col])), shape=(len(rows_l), 3 * nv))
    assert abs(Ag - Jg[rows_g]).max() <= 1e-13 * abs(Jg).max(), "owned Jacobian rows differ"

    # 3. distributed SpMV (halo-filled input) and all-reduced dot equal the serial ones
    v = rng.standard_normal(3 * nv)
    vl = np.concatenate([v[: 2 * nv].reshape(-1, 2)[l2g].ravel(), v[2 * nv:][l2g]])
    # wipe the ghosts, then refill them through the halo exchange
    vtrip = np.stack([vl[: 2 * nl].reshape(-1, 2)[:, 0], vl[: 2 * nl].reshape(-1, 2)[:, 1], vl[2 * nl:]], axis=1)
    send = torch.from_numpy(np.ascontiguousarray(vtrip[part.send_idx]).reshape(-1))
    recv = torch.zeros(3 * part.ng, dtype=torch.float64)
    comm.host_exchange(send, recv)
    vtrip[part.nvo:] = recv.numpy().reshape(-1, 3)
    vl2 = np.concatenate([vtrip[:, :2].ravel(), vtrip[:, 2]])
    y = (Jl @ vl2)[rows_l]
    assert np.abs(y - (Jg @ v)[rows_g]).max() <= 1e-12 * np.abs(Jg @ v).max()
    d = comm.allreduce(float(vl2[rows_l] @ vl2[rows_l]))
    assert abs(d - v @ v) <= 1e-12 * (v @ v)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
