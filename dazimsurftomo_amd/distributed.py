"""Sharding rules and a TEST MIRROR of the library's multi-rank algebra (torch.distributed; not a product path).

Product: everything the ranks exchange goes through the library's own communicator (csrc/comm.hip: dazim_comm_init = RCCL over
xGMI) -- the model's dispersion / TI tables sharded by rows inside dazim_dispersion_kernels_sharded / dazim_ti_kernels_sharded, the
row-sharded LSMR inside dazim_lsmr with ONE collective per iteration.  bench.py and host/dazim_main.f90 both take that path and no
other.  What lives here:

* `shard_fields`, `shard_rows`: the sharding rules (the Fortran host restates them: dazim_shard_fields / dazim_shard_rows).  The
  (source x period) eikonal fields and their rays are independent and are sharded with NO collective; G is row-partitioned as its
  rows were produced, the Tikhonov rows split evenly.
* `depthkernel_sharded`, `lsmr_distributed`: the same formulations statement by statement in torch, so that the world-size-2 / 3
  gloo tests (tests/test_distributed_cpu.py) check the algebra where no second GPU exists: blocks of model rows joined by an
  all-gather; per LSMR iteration (inv/lsmrModule.f90:36) each rank scales its shard of u by its own norm beta_p and sends
  w_p = beta_p A_p^T (u_p / beta_p) with beta_p^2 -- n fp32 + 1 fp64 in one all-gather of bytes, summed in RANK ORDER (fp32 / fp64
  like the library's k_beta_axpby); beta = sqrt(sum beta_p^2) scales afterwards; everything n-sized (v, h, hbar, x, localV) is
  replicated.  The local products come from a `LocalOps` object: `GpuLocalOps` (the HIP SpMV kernels through the C ABI) or, in
  the CPU gloo tests, a test double backed by the oracle.  The scalar recurrences are the reference's, in fp32.
"""
import numpy as np


def shard_fields(nfield, world, rank, weights=None):
    """Contiguous partition of field ids 0..nfield-1 into `world` blocks, balanced by `weights`
    (e.g. receivers per field; default 1).  Contiguity keeps the reference's period -> source ->
    receiver row order inside every shard.  Returns (start, stop)."""
    if world <= 1:
        return 0, nfield
    w = np.ones(nfield, np.float64) if weights is None else np.asarray(weights, np.float64)
    c = np.concatenate([[0.0], np.cumsum(w)])
    total = c[-1]
    bounds = [int(np.searchsorted(c, total * r / world, side="left")) for r in range(world + 1)]
    bounds[0], bounds[-1] = 0, nfield
    for r in range(1, world + 1):  # monotone, never empty while fields remain
        bounds[r] = max(bounds[r], bounds[r - 1])
    return bounds[rank], bounds[rank + 1]


def shard_rows(nrows, world, rank):
    """even contiguous split of `nrows` extra rows (Tikhonov block) -> (start, stop)"""
    base, rem = divmod(nrows, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def depthkernel_sharded(depthkernel, vel, depz, periods, minthk, world, rank, group=None, kernels=True, always_gather=False):
    """depthkernel (inv/CalSurfG.f90:1) with the model's columns sharded over the ranks: `depthkernel(vel_block, depz, periods,
    minthk, kernels=...)` -> (pv [kmax][ncol_block], sen 3 x [nz][kmax][ncol_block] or None, n_failed) is called on this rank's
    block of rows vel[:, j0:j1, :] (columns are numbered jj*nx + ii, so a block of rows is a contiguous block of columns), the
    blocks are all-gathered (padded to the largest block: all_gather wants equal shapes) and joined along the column axis.
    Every column is computed by exactly one rank with the same kernel as in the single-process call: the joined tables are
    bit-identical to it.  torch tensors (CUDA with backend nccl = RCCL, CPU with gloo).  Returns (pv, sen, n_failed summed)."""
    import torch
    import torch.distributed as dist
    nz, ny, nx = vel.shape
    if world <= 1 and not always_gather:                  # (always_gather: tests run the collective with one rank too)
        return depthkernel(vel, depz, periods, minthk, kernels=kernels)
    bounds = [shard_rows(ny, world, r) for r in range(world)]
    j0, j1 = bounds[rank]
    rows_max = max(b - a for a, b in bounds)
    kmax = len(periods)
    dev = vel.device
    if j1 > j0:
        pv_b, sen_b, nf = depthkernel(vel[:, j0:j1, :].contiguous(), depz, periods, minthk, kernels=kernels)
    else:
        pv_b, sen_b, nf = None, None, 0
    ncb = rows_max * nx
    nt = 4 if kernels else 1
    send = torch.zeros((nt, nz, kmax, ncb), dtype=torch.float64, device=dev)
    if j1 > j0:
        n = (j1 - j0) * nx
        send[0, 0, :, :n] = torch.as_tensor(pv_b, device=dev)
        if kernels:
            for q in range(3):
                send[1 + q, :, :, :n] = torch.as_tensor(sen_b[q], device=dev)
    if send.is_cuda and dist.get_backend(group) == "gloo":   # (gloo gathers host tensors only: the one-GPU rehearsal of bench.py)
        send_h = send.cpu()
        recv_h = [torch.empty_like(send_h) for _ in range(world)]
        dist.all_gather(recv_h, send_h, group=group)
        recv = [r.to(dev) for r in recv_h]
    else:
        recv = [torch.empty_like(send) for _ in range(world)]
        dist.all_gather(recv, send, group=group)
    nfail = torch.tensor([nf], dtype=torch.int64, device=dev)
    dist.all_reduce(nfail, group=group)
    cols = [(b - a) * nx for a, b in bounds]
    pv = torch.cat([recv[r][0, 0, :, :cols[r]] for r in range(world)], dim=-1).contiguous()
    sen = None
    if kernels:
        sen = [torch.cat([recv[r][1 + q, :, :, :cols[r]] for r in range(world)], dim=-1).contiguous() for q in range(3)]
    return pv, sen, int(nfail.item())


class GpuLocalOps:
    """local products on this rank's row block of G through the HIP kernels (dazim_aprod)"""

    def __init__(self, ctx, G):
        self.ctx, self.G = ctx, G
        self.m, self.n = G.m, G.n

    @staticmethod
    def _order():
        # the library launches on its own (blocking) HIP stream and returns synchronised; make the
        # torch side explicit too so that the ordering does not hinge on legacy-default-stream rules
        import torch
        torch.cuda.current_stream().synchronize()

    def aprod1(self, v, u):      # u += G_p v
        self._order()
        self.ctx.aprod(1, self.G, v, u)

    def aprod2(self, v, u):      # v += G_p^T u
        self._order()
        self.ctx.aprod(2, self.G, v, u)


def _f32(x):
    return np.float32(x)


def _d2norm(a, b):               # inv/lsmrModule.f90:708-721
    scale = _f32(abs(a) + abs(b))
    if scale == 0:
        return _f32(0)
    return _f32(scale * np.sqrt(_f32(_f32(a / scale) ** 2 + _f32(b / scale) ** 2), dtype=np.float32))


def lsmr_distributed(ops, b_local, n, damp, atol, btol, conlim, itnlim, localSize, group=None):
    """LSMR on a row-partitioned system.  `b_local` is this rank's slice of b (torch tensor on the
    device the ops work on).  Returns (x, info) with x replicated on every rank."""
    import torch
    import torch.distributed as dist

    use_dist = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    dev, f32 = b_local.device, torch.float32

    def allsum_(t):
        if use_dist:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return t

    syncs = [0]                  # device -> host scalar reads (each one waits for the device)

    def host(t):
        syncs[0] += 1
        return float(t.item())

    def gnorm_u(u):              # ||u|| over all ranks: local sum of squares in fp64, one scalar all-reduce
        s = allsum_((u.double() ** 2).sum().reshape(1))
        return _f32(np.sqrt(host(s)))

    damp, atol, btol, conlim = map(_f32, (damp, atol, btol, conlim))
    m_local = b_local.shape[0]
    m_total = int(allsum_(torch.tensor([m_local], dtype=torch.int64, device=dev)).item())
    localVecs = max(0, min(int(localSize), m_total, n))
    u = b_local.clone().to(f32)
    v = torch.zeros(n, dtype=f32, device=dev)
    x = torch.zeros(n, dtype=f32, device=dev)
    hbar = torch.zeros(n, dtype=f32, device=dev)
    info = dict(istop=0, itn=0, normA=0.0, condA=0.0, normr=0.0, normAr=0.0, normx=0.0)
    alpha = _f32(0)
    beta = gnorm_u(u)
    if beta > 0:
        u.mul_(float(_f32(1) / beta))
        ops.aprod2(v, u)
        allsum_(v)                                   # the one n-vector all-reduce of this half-step
        alpha = _f32(np.sqrt(host((v.double() ** 2).sum())))
    if alpha > 0:
        v.mul_(float(_f32(1) / alpha))
    normAr = _f32(alpha * beta)
    if normAr == 0:
        return x, info
    localV = torch.empty((max(localVecs, 1), n), dtype=f32, device=dev)
    localPointer, localVQueueFull = 0, False
    if localVecs > 0:
        localPointer = 1
        localV[0].copy_(v)
    zetabar, alphabar = _f32(alpha * beta), alpha
    rho = rhobar = cbar = _f32(1)
    sbar = _f32(0)
    h = v.clone()
    betadd, betad, rhodold, tautildeold, thetatilde, zeta, d = beta, _f32(0), _f32(1), _f32(0), _f32(0), _f32(0), _f32(0)
    normA2, maxrbar, minrbar, normb = _f32(alpha * alpha), _f32(0), _f32(1e30), beta
    ctol = _f32(1) / conlim if conlim > 0 else _f32(0)
    normr = beta
    itn = istop = 0
    normA = condA = normx = _f32(0)
    w = torch.empty(n, dtype=f32, device=dev)
    syncs[0] = 0                                      # counted over the iteration loop only
    collectives = [0]
    while True:
        itn += 1
        u.mul_(float(-alpha))
        ops.aprod1(v, u)                              # u = A_p v - alpha u   (local rows only)
        # ONE collective per iteration, as in the library (sparse.hip, k_local_norm_scal): the shard is scaled by its own norm,
        # w_p = beta_p A_p^T (u_p / beta_p), and the n values of w travel with beta_p^2 in one all-reduce
        bp2 = (u.double() ** 2).sum().reshape(1)      # local ||u_p||^2, stays on the device
        bp = torch.sqrt(bp2).to(f32)
        u.mul_(torch.where(bp > 0, 1.0 / bp, torch.ones_like(bp)))
        w.zero_()
        ops.aprod2(w, u)                              # A_p^T (u_p / beta_p)
        # ... as the library sends them (csrc/comm.hip, k_beta_axpby): the n fp32 of w_p and the double beta_p^2 in ONE all-gather of
        # bytes, then summed in RANK ORDER -- fp32 for w, fp64 for beta^2 -- so that this mirror rounds where the product rounds
        pack = torch.cat([(w * bp).contiguous().view(torch.uint8), bp2.contiguous().view(torch.uint8)])
        if use_dist:
            parts = [torch.empty_like(pack) for _ in range(dist.get_world_size(group))]
            dist.all_gather(parts, pack, group=group)
        else:
            parts = [pack]
        collectives[0] += 1
        wsum = parts[0][:4 * n].view(f32).clone()
        b2sum = parts[0][4 * n:].view(torch.float64).clone()
        for q in parts[1:]:
            wsum = wsum + q[:4 * n].view(f32)
            b2sum = b2sum + q[4 * n:].view(torch.float64)
        beta = _f32(np.sqrt(host(b2sum[0])))
        if beta > 0:
            u.mul_(bp * float(_f32(1) / beta))        # u_p = u / beta
            if localVecs > 0:                         # localVEnqueue
                if localPointer < localVecs:
                    localPointer += 1
                else:
                    localPointer, localVQueueFull = 1, True
                localV[localPointer - 1].copy_(v)
            v.mul_(float(-beta)).add_(wsum * float(_f32(1) / beta))
            if localVecs > 0:                         # localVOrtho, modified Gram-Schmidt
                lim = localVecs if localVQueueFull else localPointer
                for q in range(lim):
                    dq = torch.dot(v, localV[q])
                    v.sub_(localV[q] * dq)
            alpha = _f32(np.sqrt(host((v.double() ** 2).sum())))
            if alpha > 0:
                v.mul_(float(_f32(1) / alpha))
        alphahat = _d2norm(alphabar, damp)
        chat, shat = _f32(alphabar / alphahat), _f32(damp / alphahat)
        rhoold = rho
        rho = _d2norm(alphahat, beta)
        c, s = _f32(alphahat / rho), _f32(beta / rho)
        thetanew = _f32(s * alpha)
        alphabar = _f32(c * alpha)
        rhobarold, zetaold = rhobar, zeta
        thetabar, rhotemp = _f32(sbar * rho), _f32(cbar * rho)
        rhobar = _d2norm(_f32(cbar * rho), thetanew)
        cbar = _f32(_f32(cbar * rho) / rhobar)
        sbar = _f32(thetanew / rhobar)
        zeta = _f32(cbar * zetabar)
        zetabar = _f32(-sbar * zetabar)
        f1 = float(_f32(_f32(thetabar * rho) / _f32(rhoold * rhobarold)))
        f2 = float(_f32(zeta / _f32(rho * rhobar)))
        f3 = float(_f32(thetanew / rho))
        hbar.mul_(-f1).add_(h)
        x.add_(hbar, alpha=f2)
        h.mul_(-f3).add_(v)
        betaacute, betacheck = _f32(chat * betadd), _f32(-shat * betadd)
        betahat = _f32(c * betaacute)
        betadd = _f32(-s * betaacute)
        thetatildeold = thetatilde
        rhotildeold = _d2norm(rhodold, thetabar)
        ctildeold, stildeold = _f32(rhodold / rhotildeold), _f32(thetabar / rhotildeold)
        thetatilde = _f32(stildeold * rhobar)
        rhodold = _f32(ctildeold * rhobar)
        betad = _f32(_f32(-stildeold * betad) + _f32(ctildeold * betahat))
        tautildeold = _f32(_f32(zetaold - _f32(thetatildeold * tautildeold)) / rhotildeold)
        taud = _f32(_f32(zeta - _f32(thetatilde * tautildeold)) / rhodold)
        d = _f32(d + _f32(betacheck * betacheck))
        normr = _f32(np.sqrt(_f32(_f32(d + _f32(_f32(betad - taud) ** 2)) + _f32(betadd * betadd)), dtype=np.float32))
        normA2 = _f32(normA2 + _f32(beta * beta))
        normA = _f32(np.sqrt(normA2, dtype=np.float32))
        normA2 = _f32(normA2 + _f32(alpha * alpha))
        maxrbar = max(maxrbar, rhobarold)
        if itn > 1:
            minrbar = min(minrbar, rhobarold)
        condA = _f32(max(maxrbar, rhotemp) / min(minrbar, rhotemp))
        normAr = _f32(abs(zetabar))
        normx = _f32(np.sqrt(host((x.double() ** 2).sum())))
        test1 = _f32(normr / normb)
        test2 = _f32(normAr / _f32(normA * normr))
        test3 = _f32(_f32(1) / condA)
        t1 = _f32(test1 / _f32(_f32(1) + _f32(_f32(normA * normx) / normb)))
        rtol = _f32(btol + _f32(_f32(_f32(atol * normA) * normx) / normb))
        if itn >= itnlim: istop = 7
        if _f32(1) + test3 <= 1: istop = 6
        if _f32(1) + test2 <= 1: istop = 5
        if _f32(1) + t1 <= 1: istop = 4
        if test3 <= ctol: istop = 3
        if test2 <= atol: istop = 2
        if test1 <= rtol: istop = 1
        if istop:
            break
    if damp > 0 and istop == 2:
        istop = 3
    info.update(istop=int(istop), itn=itn, normA=float(normA), condA=float(condA), normr=float(normr),
                normAr=float(normAr), normx=float(normx), host_syncs=syncs[0], collectives=collectives[0])
    return x, info
