"""Worker of the multi-process tests (launched by torch.distributed.run, gloo rendezvous on 127.0.0.1).

mode "lists": CPU only.  Every rank builds its shard of the Schwarz operator in the product library (host side:
Subdomain::initialize sorting, SetPartition, cross-GPU halo lists), then the halo sum is replayed in numpy with the
library's lists -- pack, gloo send/recv with the library's peer layout, unpack -- and compared with the oracle's global
exchange.  This pins the ordering contract of the N>1 path without a GPU.

mode "gpu": all ranks share GPU 0 (host-staged gloo transport).  apply / GMV / GMRES of the sharded operator are
compared with the oracle on the global problem.

mode "rccl": one GPU per rank, the library's own RCCL transport (HpddmHipSchwarzInitRccl; the ncclUniqueId travels over the
gloo group), same checks as "gpu".  Needs as many GPUs as ranks.

Layout (every mode): the topology BASELINE.json's multi-GPU configs name -- every rank owns one 2 x 2 x 2 brick of subdomains,
the bricks form the most cubic grid of `world` GPUs (8 ranks: 4 x 4 x 4 subdomains as 2 x 2 x 2 bricks, every GPU a neighbour of the
7 others = configs[3]; 4 ranks: 2 x 4 x 4 subdomains = the layout of configs[4]; 2 / 3 ranks: a chain).  Subdomains are numbered
brick by brick, so the contiguous ranges of HpddmHipSchwarzSetPartition are the bricks.

Optional second argument: "helmholtz" -- complex<double> operator (shifted Laplacian with absorption), Block GMRES on 2 right-hand sides.
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from hpddm_amd import hpddm  # noqa: E402
from hpddm_amd.generate import generate3d, generate_helmholtz3d, gpu_grid  # noqa: E402
from oracle.ras_oracle import Oracle  # noqa: E402


def main():
    mode = sys.argv[1]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    helm = len(sys.argv) > 2 and sys.argv[2] == "helmholtz"
    gg = gpu_grid(world)                                   # bricks (= GPUs) per direction
    grid = tuple(2 * g for g in gg)                        # subdomains per direction
    cells = 3 if world >= 8 else 4                         # cells per subdomain and direction, before the overlap
    parts, per, dims = 8 * world, 8, tuple(cells * g for g in grid)
    firsts = [r * per for r in range(world + 1)]
    if helm:   # the problem of configs[4] (absorbing boundary, ORAS impedance matrices, DtN pencil), k h = 0.52
        allsubs = generate_helmholtz3d(dims, parts, wavenumber=0.52 * max(dims), grid=grid, brick=(2, 2, 2))
        for sd in allsubs:
            sd["f"] = np.ones(sd["n"])
    else:
        allsubs = generate3d(dims, parts, overlap=1, sym=True, rhs="smooth", grid=grid, brick=(2, 2, 2), normalize=True)
    mine = allsubs[firsts[rank]:firsts[rank + 1]]
    orc = Oracle(allsubs)
    orc.d = [s["d"] for s in allsubs]
    A, d = hpddm.schwarz_from_subdomains(mine, first_global=firsts[rank], nglobal=parts, options="-hpddm_schwarz_method oras" if helm else "-hpddm_operator_spd", multiplicity=False,
                                         partition=(rank, firsts))
    if helm:
        orc.method = "oras"
        for k, sd in enumerate(mine):
            A.set_optimized_matrix(k, sd["n"], sd["ia"], sd["ja"], sd["a_opt"], False)
    rng = np.random.default_rng(5)
    mu = 2
    xg = [rng.random((s["n"], mu)) + (1j * rng.random((s["n"], mu)) if helm else 0.0) for s in allsubs]
    ref = orc.exchange(xg)
    peers = A.halo_peers()
    # the peers of a brick: every brick within distance 1 in each direction (7 xGMI links per GPU on the 2 x 2 x 2 grid of configs[3])
    me = (rank % gg[0], rank // gg[0] % gg[1], rank // (gg[0] * gg[1]))
    expect = sorted(x + gg[0] * (y + gg[1] * z) for x in range(gg[0]) for y in range(gg[1]) for z in range(gg[2])
                    if (x, y, z) != me and max(abs(x - me[0]), abs(y - me[1]), abs(z - me[2])) <= 1)
    assert [p for p, _, _ in peers] == expect, (peers, expect)
    if world == 8:
        assert len(peers) == 7
    # the ordering contract of every link, checked on both of its ends: what a sends to b, block by block (source subdomain,
    # destination subdomain, dofs), is what b expects from a
    sp, rp = A.halo_export("send_pairs").reshape(-1, 4), A.halo_export("recv_pairs").reshape(-1, 4)
    box = [None] * world
    dist.all_gather_object(box, (sp.tolist(), rp.tolist()))
    for a_ in range(world):
        for b_ in range(world):
            sent = [q[1:] for q in box[a_][0] if q[0] == b_]
            want = [q[1:] for q in box[b_][1] if q[0] == a_]
            assert sent == want, (a_, b_, sent[:4], want[:4])
            assert all(s // per == a_ and t // per == b_ for s, t, _ in sent)
            assert sent == sorted(sent), "blocks of a message in increasing (source, destination) order"
    assert sum(q[3] for q in sp.tolist()) == sum(c for _, c, _ in peers) == sum(q[3] for q in rp.tolist())
    if mode == "lists_rccl":
        # CPU, no device: the product's RcclTransport -- its grouped ncclSend / ncclRecv sequence with the peer offsets and counts of
        # the partition, ncclSum / ncclMax -- against tests/fake_rccl (a double of librccl over /dev/shm that refuses a message whose
        # size differs from what the receiving end expects); every entry of the send buffer is tagged with (rank, position), so the
        # receive buffer says where every value came from
        assert os.environ.get("FAKE_RCCL_HOST") == "1" and "fake_rccl" in os.environ.get("HPDDM_HIP_RCCL_LIB", "")
        total = sum(c for _, c, _ in peers)
        send = 1e6 * rank + np.arange(total * mu, dtype=np.float64)
        box = [hpddm.rccl_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        rs, rm = np.array([rank + 1.0, 0.5, -rank]), np.array([rank + 1.0, 0.5, -rank])
        recv = hpddm.rccl_halo_probe(A, box[0], send, mu, rs, rm)
        assert np.array_equal(rs, [world * (world + 1) / 2, 0.5 * world, -world * (world - 1) / 2]) and np.array_equal(rm, [world, 0.5, 0.0]), (rs, rm)
        layout = [None] * world
        dist.all_gather_object(layout, peers)
        for prank, cnt, off in peers:
            # the block of peer `prank` in my receive buffer = the block it keeps for me in its send buffer, value by value
            poff = [o for q, c, o in layout[prank] if q == rank]
            pcnt = [c for q, c, o in layout[prank] if q == rank]
            assert len(poff) == 1 and pcnt[0] == cnt, (rank, prank, cnt, pcnt)
            want = 1e6 * prank + np.arange(poff[0] * mu, (poff[0] + cnt) * mu, dtype=np.float64)
            assert np.array_equal(recv[off * mu:(off + cnt) * mu], want), (rank, prank)
        dist.barrier()
        if rank == 0:
            print(f"DIST_WORKER_OK mode={mode} world={world} peers={peers}")
        dist.destroy_process_group()
        return
    if mode == "lists":
        assert not helm
        L = {k: A.halo_export(k) for k in ("send_sub", "send_idx", "send_po", "send_pc", "rx_ptr", "rx_k", "rx_po", "rx_pc")}
        total = sum(c for _, c, _ in peers)
        assert len(L["send_sub"]) == total
        voff = np.concatenate([[0], np.cumsum([s["n"] for s in mine])])
        x = xg[firsts[rank]:firsts[rank + 1]]
        # local part (what k_exchange does): D x + co-located neighbours in neighbour order
        out = [d[s][:, None] * x[s] for s in range(per)]
        sc = [o.copy() for o in out]
        for s, sd in enumerate(mine):
            order = np.argsort(sd["neighbors"], kind="stable")
            for k in order:
                t = int(sd["neighbors"][k]) - firsts[rank]
                if 0 <= t < per:
                    kt = list(mine[t]["neighbors"]).index(firsts[rank] + s)
                    out[s][sd["connectivity"][k]] += sc[t][mine[t]["connectivity"][kt]]
        # remote part with the library's lists: pack
        send = np.zeros(total * mu)
        for k in range(total):
            s, i, po, pc = L["send_sub"][k], L["send_idx"][k], L["send_po"][k], L["send_pc"][k]
            for nu in range(mu):
                send[po * mu + nu * pc + (k - po)] = d[s][i] * x[s][i, nu]
        recv = np.zeros_like(send)
        ops, bufs = [], []
        for prank, cnt, off in peers:
            sb = torch.from_numpy(send[off * mu:(off + cnt) * mu].copy())
            rb = torch.zeros(cnt * mu, dtype=torch.float64)
            bufs.append((off, cnt, rb))
            ops += [dist.P2POp(dist.isend, sb, prank), dist.P2POp(dist.irecv, rb, prank)]
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        for off, cnt, rb in bufs:
            recv[off * mu:(off + cnt) * mu] = rb.numpy()
        # unpack
        for s in range(per):
            for i in range(mine[s]["n"]):
                g = voff[s] + i
                for p in range(L["rx_ptr"][g], L["rx_ptr"][g + 1]):
                    k, po, pc = L["rx_k"][p], L["rx_po"][p], L["rx_pc"][p]
                    for nu in range(mu):
                        out[s][i, nu] += recv[po * mu + nu * pc + (k - po)]
        err = max(np.abs(o - r).max() for o, r in zip(out, ref[firsts[rank]:firsts[rank + 1]]))
        assert err < 1e-14, err
    else:
        hpddm.require_device()
        if mode == "rccl":
            from hpddm_amd import _lib
            local = int(os.environ.get("LOCAL_RANK", rank))
            if os.environ.get("HPDDM_TEST_RCCL_SAME_GPU") == "1":   # probe only: RCCL refuses two ranks of a communicator on one device
                local = 0
            assert hpddm.device_count() > local, "mode rccl needs one GPU per rank"
            _lib.check(_lib.load().HpddmHipSetDevice(local))
            box = [hpddm.rccl_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            A.enable_rccl(box[0], mu_cap=4)
        else:
            dev = torch.device("cuda", 0)
            torch.cuda.set_device(dev)
            A.enable_distributed(dist, dev, mu_cap=4, host_staging=True)
        A.call_numfact()
        if helm:
            import scipy.sparse as sp
            orc.numfact([sp.csr_matrix((sd["a_opt"], sd["ja"], sd["ia"]), shape=(sd["n"], sd["n"])) for sd in allsubs])
        else:
            orc.numfact()
        x = xg[firsts[rank]:firsts[rank + 1]]
        sl = slice(firsts[rank], firsts[rank + 1])

        def close(a, b, tol, what):
            scale = max(np.abs(v).max() for v in b)
            err = max(np.abs(u - v).max() for u, v in zip(a, b)) / scale
            assert err < tol, (what, err)

        close(A.exchange(x), ref[sl], 1e-14, "exchange")
        close(A.gmv(x), orc.gmv(xg)[sl], 1e-13, "gmv")
        f = orc.exchange(xg)
        close(A.apply(f[sl]), orc.apply(f)[sl], 1e-10, "apply")
        if helm:
            # configs[4] in small: complex operator, plane-wave coarse space assembled across the ranks, Block GMRES on the block of
            # right-hand sides (the Gram matrices of the block method are summed over the ranks)
            from oracle import ras_oracle as ro
            # DtN coarse space: the complex solveGEVP(A_Neumann, B_interface) of every local subdomain on the device; the oracle takes
            # ARPACK's vectors of the same pencils (the operators do not depend on the basis: nu = 4 cuts no cluster here -- checked below)
            nu = 4
            A.set_option("geneo_nu", nu)
            A.set_option("eigensolver_tol", 1e-11)
            mats = lambda key: [sp.csr_matrix((sd[key], sd["ja"], sd["ia"]), shape=(sd["n"], sd["n"])) for sd in allsubs]
            Bs = [sp.csr_matrix((sd["b_dtn"][2], sd["b_dtn"][1], sd["b_dtn"][0]), shape=(sd["n"], sd["n"])) for sd in allsubs]
            lam_ref = orc.geneo_z(mats("a_neumann"), nu + 1, B=Bs)
            orc.set_vectors([z[:, :nu] for z in orc.Z])
            for k, sd in enumerate(mine):
                lam = A.solve_gevp(k, sd["n"], sd["ia"], sd["ja"], sd["a_neumann"], False, B=sd["b_dtn"] + (False,))
                ref = lam_ref[firsts[rank] + k]
                assert np.all(np.abs(lam - ref[:nu]) <= 1e-6 * np.abs(ref[:nu])) and abs(ref[nu]) > (1 + 1e-3) * abs(ref[nu - 1]), (lam, ref)
            A.build_coarse_operator()
            orc.build_coarse(lapacktr=False)
            A.option_parse("-hpddm_schwarz_coarse_correction deflated -hpddm_krylov_method bgmres -hpddm_gmres_restart 20")
            orc.correction = "deflated"
            close(A.deflation(f[sl]), orc.deflation(f)[sl], 1e-9, "deflation")
            close(A.apply(f[sl]), orc.apply(f)[sl], 1e-9, "apply deflated")
            it, sol = A.solve(f[sl])
            it_o, sol_o, _ = ro.bgmres(orc, f, restart=20)
            assert it == it_o, (it, it_o)
            close(sol, sol_o[sl], 1e-7, "solution")
            dist.barrier()
            if rank == 0:
                print(f"DIST_WORKER_OK mode={mode} helmholtz world={world} peers={peers} bgmres={it}")
            dist.destroy_process_group()
            return
        it, sol = A.solve(f[sl])
        it_o, sol_o, _ = orc.gmres(f)
        assert it == it_o, (it, it_o)
        close(sol, sol_o[sl], 1e-8, "solution")
        res = A.compute_residual(sol, f[sl])
        assert np.allclose(res, orc.compute_residual(sol_o, f), rtol=1e-4), res
        # l1 (sum over the ranks) and linfty (MPI_MAX in the reference, include/HPDDM_schwarz.hpp:802: a max-reduction of the transport)
        for nrm in ("l1", "linfty"):
            res = A.compute_residual(x, f[sl], norm=nrm)
            assert np.allclose(res, orc.compute_residual(xg, f, norm=nrm), rtol=1e-10), (nrm, res)
        # two-level: the coarse operator is assembled across the ranks (neighbours' A D Z fetched through the halo
        # transport, more columns than the transport's mu_cap -> chunked), E^{-1} replicated, coarse gather = all-reduce
        zr = np.random.default_rng(11)
        Zg = [np.column_stack([np.ones(s["n"])] + [zr.random(s["n"]) for _ in range(4 + k % 3)]) for k, s in enumerate(allsubs)]
        for k in range(per):
            A.set_vectors(k, Zg[firsts[rank] + k])
        A.build_coarse_operator()
        orc.set_vectors(Zg)
        orc.build_coarse()
        close(A.deflation(f[sl]), orc.deflation(f)[sl], 1e-9, "deflation")
        for corr in ("deflated", "additive", "balanced"):
            A.option_parse("-hpddm_schwarz_coarse_correction " + corr)
            orc.correction = corr
            close(A.apply(f[sl]), orc.apply(f)[sl], 1e-9, "apply " + corr)
        it2, sol2 = A.solve(f[sl])
        it2_o, sol2_o, _ = orc.gmres(f)
        assert it2 == it2_o and it2 < it, (it2, it2_o, it)
        close(sol2, sol2_o[sl], 1e-8, "two-level solution")
    dist.barrier()
    if rank == 0:
        print(f"DIST_WORKER_OK mode={mode} world={world} peers={peers}")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
