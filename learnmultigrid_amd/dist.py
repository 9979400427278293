"""Multi-GPU V-cycle: contiguous row-block partition of every level, halo exchange over
RCCL (torch.distributed backend "nccl" IS RCCL on ROCm), residual-norm all-reduce.

The reference is single-process (no MPI / NCCL anywhere); this layer is new work defined
by BASELINE.json's north star, designed for one node of 8 x MI355X (288 GB each,
point-to-point xGMI links):

  * SETUP IS REPLICATED.  Every rank builds the whole hierarchy on its own GPU (the
    4097^2 hierarchy is ~3 GB, the 8193^2 one ~12 GB, of 288 GB) and slices its row
    blocks out of it: no distributed SpGEMM, no setup-time collectives except one
    all_gather_object of the ghost index lists.
  * Level l is owned in contiguous row blocks; with `grid_side` the cuts fall on grid
    lines and coarse line j lives with fine line 2j, so A, R = P^T and P all need ghosts
    from the two neighbouring ranks only.  Each level has ONE ghost set (union of what
    A_l, R_l and P_{l-1} reference), so every level-l vector has the layout
    [owned rows | ghosts sorted by global index] and one exchange plan.
  * One halo exchange = a pack kernel (lmg_gather) + one grouped batch of
    isend / irecv (<= 2 neighbours).  Messages are latency-bound, not xGMI-bandwidth-bound,
    so the cycle is organised to need FEW of them: the ghost set of a level reaches
    `halo_depth` (default 8) matrix hops beyond the owned block and the local operators A_l
    and P_l carry the real rows of all ghost layers but the outermost.  After one exchange of
    x every sweep on the whole local block is exact one layer less deep than the previous
    one, so nu sweeps, the residual on the owned rows AND on the ghost layers the restriction
    reads (two hops of a 5-point operator for a 9-point restriction) need no further message
    (halo_depth >= nu + 3).  The correction is prolongated on the ghost layers as well, so the
    corrected iterate is still exact on halo_depth - nu >= nu layers and post-smoothing needs
    no message either (halo_depth >= 2 nu); and what post-smoothing leaves exact on a coarse
    level (halo_depth - 2 nu layers) is enough for the finer level's prolongation when
    halo_depth >= 2 nu + 2.  With nu = 3 and depth 8 that is ONE exchange per distributed level
    and cycle -- x on the fine level, b on the coarse ones (their iterate starts from zero) --
    instead of 2 nu + 2; cfg#4: 2 instead of 16.  The redundant work is halo_depth grid lines
    per neighbour (262 KB messages on the fine level of cfg#4).  All depths are measured at
    setup from the hop distances of the columns of R and P and agreed over the ranks; levels
    or cycles that need more than the halo offers fall back, per level, to one exchange per use.
  * Levels with fewer than `replicate_below` rows (default 2 M: below that a halo
    exchange costs more than computing the whole level redundantly) are NOT distributed: the restricted
    residual is all-gathered once per cycle and every rank runs the rest of the cycle
    redundantly on its replicated hierarchy, then prolongates from the full coarse
    vector without further communication.
  * Weighted Jacobi: row sums are accumulated in the same storage order as on one
    GPU, so the distributed iterate is bit-identical to the single-GPU one; only the
    all-reduced norm differs in summation order.
  * Gauss-Seidel -- the smoother the reference ships (Multigrid.py:88, :121) -- as PROCESSOR-BLOCK
    ("hybrid") Gauss-Seidel: lexicographic order is sequential across the partition, so on a
    distributed level every rank relaxes ITS rows in lexicographic order with the ghost values of
    the start of the sweep (one halo exchange per sweep; exact forward sweeps inside the block:
    the wavefront / level-scheduled kernels on the owned rows).  With one rank it is the
    reference's sweep; with P ranks it is the standard block variant, and it has its own CPU twin
    (oracle/vcycle_ref.py HybridGSVCycle) against which it is bit-identical.  The replicated
    levels below run the exact sweep on every rank.

The class is written against an `ops` object (default: the HIP kernels) so that the
partition / halo / collective logic is exercised by world_size-2 gloo tests on CPU
tensors with a test-only ops shim; the product never falls back to it.
"""
import math

import numpy as np
import torch
import torch.distributed as dist

from .ops import DeviceCSR, F64, I32


def block_bounds(n_units, world):
    """Contiguous split of n_units into `world` blocks, sizes differing by at most 1."""
    base, rem = divmod(n_units, world)
    b = [0]
    for p in range(world):
        b.append(b[-1] + base + (1 if p < rem else 0))
    return b


def _rows(M, lo, hi):
    """Rows [lo, hi) of a CSR held in tensors: (rowptr_local, colidx_global, vals, source) --
    `source` says where the values came from (see _take)."""
    rp = M.rowptr[lo:hi + 1].long()
    s, e = int(rp[0]), int(rp[-1])
    return (rp - s).to(I32), M.colidx[s:e], M.vals[s:e], ("slice", s, e)


def _gather_rows(M, rows, with_source=False):
    """Entries of the given rows (int64 tensor) of a CSR in tensors: (counts, colidx, vals),
    rows in the given order, entries in storage order."""
    if rows.numel() == 0:
        out = (torch.zeros(0, dtype=torch.long, device=M.vals.device), M.colidx[:0], M.vals[:0])
        return out + (("slice", 0, 0),) if with_source else out
    start = M.rowptr[rows].long()
    cnt = M.rowptr[rows + 1].long() - start
    tot = int(cnt.sum())
    if tot == 0:
        out = (cnt, M.colidx[:0], M.vals[:0])
        return out + (("slice", 0, 0),) if with_source else out
    first = torch.cumsum(cnt, 0) - cnt                       # position of every row's first entry
    idx = torch.repeat_interleave(start - first, cnt) + torch.arange(tot, device=M.vals.device)
    out = (cnt, M.colidx[idx], M.vals[idx])
    return out + (("index", idx),) if with_source else out


def _take(vals, sources):
    """Values of a local operator from the replicated one, by the sources recorded when it was cut."""
    return torch.cat([vals[src[1]:src[2]] if src[0] == "slice" else vals[src[1]] for src in sources])


class _DLevel:
    """One distributed level on this rank.  Vector layout (length n_tot):
        [ghosts below lo | owned rows lo..hi | ghosts above hi]      (all in global order)
    Local matrices have n_tot rows too -- ghost rows are EMPTY -- so row index == column index
    on the diagonal (the kernels find the diagonal that way), every kernel takes whole
    vectors, and local column indices grow with the global ones (tile column spans stay small:
    the packed 16-bit column encoding keeps working)."""

    def to_local(self, c):
        c = c.long()
        below = torch.searchsorted(self.ghost_lo, c)
        above = torch.searchsorted(self.ghost_hi, c)
        return torch.where(c < self.lo, below,
                           torch.where(c < self.hi, c - self.lo + self.n_lo, self.n_lo + self.n_own + above))

    def embed_rows(self, rp_owned):
        """rowptr of the owned rows -> rowptr over the n_tot layout rows (ghost rows empty)."""
        nnz = rp_owned[-1:]
        return torch.cat([torch.zeros(self.n_lo, dtype=I32, device=rp_owned.device), rp_owned,
                          nnz.expand(self.n_hi)]).contiguous()


class DistributedVCycle:
    def __init__(self, full, device, ops_mod=None, grid_side=None, replicate_below=2_000_000,
                 group=None, halo_depth=8):
        """full: a replicated hierarchy (learnmultigrid_amd.hierarchy.Hierarchy or anything
        with the same .levels[l].A/.P/.R, .coarse_solve(), .cycle())."""
        if ops_mod is None:
            from . import ops as ops_mod
        self.ops = ops_mod
        self.full = full
        self.device = torch.device(device)
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.stream = getattr(full, "stream", None)
        # "gloo" cannot move device tensors point to point: when it drives GPU ranks (tests that put
        # several ranks on one GPU, or a node without RCCL) messages are staged through the host
        self.host_staged = self.device.type == "cuda" and dist.get_backend(group) == "gloo"
        # ghost layers per level (matrix hops); cycles with nu + 2 <= halo_depth exchange 3 times
        # per level, others before every sweep.  1 = the classic one-layer halo.
        self.halo_depth = max(1, int(halo_depth))
        self.n_exchanges = 0                      # halo exchanges issued so far (tests, bench)
        nlev = len(full.levels)
        sizes = [lev.n for lev in full.levels]
        # ---- which levels are distributed -------------------------------------------------
        self.n_dist = 0
        for l in range(nlev - 1):                   # the coarsest level is always replicated
            if sizes[l] >= replicate_below and sizes[l] >= 4 * self.world:
                self.n_dist = l + 1
            else:
                break
        if self.n_dist == 0:
            raise ValueError("problem too small to distribute (fine level has %d rows)" % sizes[0])
        # ---- row ranges per level ------------------------------------------------------------
        self.bounds = []
        side = grid_side
        for l in range(self.n_dist + 1):
            n = sizes[l]
            if l == 0:
                if side is not None and side * side == n:
                    cuts = [c * side for c in block_bounds(side, self.world)]
                else:
                    side = None
                    cuts = block_bounds(n, self.world)
            else:
                # coarse row c lives with the rank that owns its anchor fine row (the row of
                # the largest entry of column c of P): coarse line j stays with fine line 2j
                R = full.levels[l - 1].R
                anchor = self._anchors(R)
                prev = torch.tensor(self.bounds[l - 1], device=anchor.device, dtype=anchor.dtype)
                cm = torch.cummax(anchor, 0).values
                cuts = torch.searchsorted(cm, prev[:-1].contiguous()).tolist() + [n]
                cuts[0] = 0
            self.bounds.append([int(c) for c in cuts])
        # ---- local slices + ghost sets ---------------------------------------------------------
        raw = []
        for l in range(self.n_dist):
            lo, hi = self.bounds[l][self.rank], self.bounds[l][self.rank + 1]
            clo, chi = self.bounds[l + 1][self.rank], self.bounds[l + 1][self.rank + 1]
            lev = full.levels[l]
            raw.append({"A": _rows(lev.A, lo, hi), "P": _rows(lev.P, lo, hi), "R": _rows(lev.R, clo, chi)})
        # The prolongation is applied on every ghost layer that carries real rows too, so that the
        # corrected iterate needs no exchange before post-smoothing (see cycle()).
        ghosts, real_ghosts, r_need, p_ghost_rows, layers_all = [], [], [], [], []
        for l in range(self.n_dist):
            lo, hi = self.bounds[l][self.rank], self.bounds[l][self.rank + 1]
            A_l = full.levels[l].A

            def outside(c, known):
                c = c.long()
                c = torch.unique(c[(c < lo) | (c >= hi)])
                if known.numel():
                    pos = torch.searchsorted(known, c).clamp(max=known.numel() - 1)
                    c = c[known[pos] != c]
                return c

            # layer k = rows reached from the owned block in k hops through A_l; the rows of all
            # layers but the last are kept as REAL rows of the local operator
            known = torch.zeros(0, dtype=torch.long, device=A_l.vals.device)
            frontier = outside(raw[l]["A"][1], known)
            real = known
            layers = []
            for k in range(self.halo_depth):
                layers.append(frontier)
                known = torch.unique(torch.cat([known, frontier]))
                if k + 1 == self.halo_depth or frontier.numel() == 0:
                    break
                real = known
                frontier = outside(_gather_rows(A_l, frontier)[1], known)
            # deepest ghost layer the restriction reads (a residual is exact on layer k only if x
            # is exact on layer k + 1); columns that A does not reach within halo_depth hops at all
            # rule the few-exchanges scheme out on this level
            rc = outside(raw[l]["R"][1], known[:0])
            need = 0
            if rc.numel():
                lay = torch.full((rc.numel(),), 1 << 20, dtype=torch.long, device=rc.device)
                for k, fr in enumerate(layers, 1):
                    if fr.numel():
                        pos = torch.searchsorted(fr, rc).clamp(max=fr.numel() - 1)
                        lay = torch.where(fr[pos] == rc, torch.full_like(lay, k), lay)
                need = int(lay.max())
            r_need.append(need)
            p_ghost_rows.append(real)                  # ghost rows of this level that carry real P rows
            layers_all.append(layers)
            extra = [raw[l]["R"][1]]
            if l > 0:
                extra.append(raw[l - 1]["P"][1])
                # ... and what the P rows of the finer level's inner ghost layers reference
                extra.append(_gather_rows(full.levels[l - 1].P, p_ghost_rows[l - 1])[1])
            known = torch.unique(torch.cat([known, outside(torch.cat(extra), known)]))
            ghosts.append(known)                       # sorted
            real_ghosts.append(real)
        # p_need[l][k]: deepest ghost layer of level l+1 that the P rows of the level-l ghost layers
        # <= k read (0: owned coarse rows only; the replicated level below the last distributed one
        # is complete on every rank)
        p_need = []
        for l in range(self.n_dist):
            need_k = [0] * (self.halo_depth + 1)
            if l + 1 < self.n_dist:
                clo, chi = self.bounds[l + 1][self.rank], self.bounds[l + 1][self.rank + 1]
                worst = 0
                for k in range(0, self.halo_depth):
                    rows_k = None if k == 0 else (layers_all[l][k - 1] if k - 1 < len(layers_all[l]) else None)
                    cols = raw[l]["P"][1] if k == 0 else (
                        _gather_rows(full.levels[l].P, rows_k)[1] if rows_k is not None and rows_k.numel() else None)
                    if cols is not None and cols.numel():
                        c = torch.unique(cols.long())
                        c = c[(c < clo) | (c >= chi)]
                        if c.numel():
                            lay = torch.full((c.numel(),), 1 << 20, dtype=torch.long, device=c.device)
                            for kk, fr in enumerate(layers_all[l + 1], 1):
                                if fr.numel():
                                    pos = torch.searchsorted(fr, c).clamp(max=fr.numel() - 1)
                                    lay = torch.where(fr[pos] == c, torch.full_like(lay, kk), lay)
                            worst = max(worst, int(lay.max()))
                    need_k[k] = worst
                need_k[self.halo_depth] = worst
            p_need.append(need_k)
        gathered = [None] * self.world
        dist.all_gather_object(gathered, [g.cpu().numpy() for g in ghosts] + [r_need, p_need], group=group)
        # every rank must take the same branch of the cycle (the exchanges are collective)
        self.r_need = [max(g[-2][l] for g in gathered) for l in range(self.n_dist)]
        self.p_need = [[max(g[-1][l][k] for g in gathered) for k in range(self.halo_depth + 1)]
                       for l in range(self.n_dist)]
        self.dl = []
        for l in range(self.n_dist):
            d = _DLevel()
            lo, hi = self.bounds[l][self.rank], self.bounds[l][self.rank + 1]
            d.lo, d.hi, d.n_own = lo, hi, hi - lo
            g = ghosts[l]
            d.ghost_lo, d.ghost_hi = g[g < lo].contiguous(), g[g >= hi].contiguous()
            d.n_lo, d.n_hi = int(d.ghost_lo.numel()), int(d.ghost_hi.numel())
            d.n_tot = d.n_lo + d.n_own + d.n_hi
            d.own = slice(d.n_lo, d.n_lo + d.n_own)
            # exchange plan: per source rank one contiguous ghost segment (ghosts are sorted and
            # ownership is contiguous)
            gb = torch.tensor(self.bounds[l], device=g.device)
            d.recv = []
            for part, base in ((d.ghost_lo, 0), (d.ghost_hi, d.n_lo + d.n_own)):
                if part.numel() == 0:
                    continue
                owner = torch.searchsorted(gb, part, right=True) - 1
                for q in torch.unique(owner).tolist():
                    idx = torch.nonzero(owner == q).flatten()
                    d.recv.append((int(q), base + int(idx[0]), int(idx.numel())))
            d.send = []
            for q in range(self.world):
                if q == self.rank:
                    continue
                want = gathered[q][l]
                mine = want[(want >= lo) & (want < hi)] - lo + d.n_lo
                if mine.size:
                    if mine.size == int(mine[-1]) - int(mine[0]) + 1:
                        # structured row blocks: the boundary grid line is one contiguous run,
                        # sent straight out of the vector (no pack kernel, no staging buffer)
                        d.send.append((q, (int(mine[0]), int(mine[-1]) + 1), None))
                    else:
                        d.send.append((q, torch.from_numpy(mine.astype(np.int32)).to(self.device),
                                       torch.empty(mine.size, dtype=F64, device=self.device)))
            for name in ("x", "b", "r", "tmp"):
                setattr(d, name, torch.zeros(d.n_tot, dtype=F64, device=self.device))
            self.dl.append(d)
        # ---- the exchange plans of all ranks must pair up: a send without its receive would hang in RCCL, so
        # check it here, once, where a mismatch can still raise ---------------------------------------------
        plan = [([(q, (idx[1] - idx[0]) if isinstance(idx, tuple) else int(idx.numel())) for q, idx, _b in d.send],
                 [(q, cnt) for q, _off, cnt in d.recv]) for d in self.dl]
        plans = [None] * self.world
        dist.all_gather_object(plans, plan, group=group)
        for l in range(self.n_dist):
            for p in range(self.world):
                for q, cnt in plans[p][l][0]:
                    if (p, cnt) not in plans[q][l][1]:
                        raise RuntimeError("halo plan of level %d: rank %d sends %d values to rank %d, which does not "
                                           "expect them" % (l, p, cnt, q))
                for q, cnt in plans[p][l][1]:
                    if (p, cnt) not in plans[q][l][0]:
                        raise RuntimeError("halo plan of level %d: rank %d expects %d values from rank %d, which does "
                                           "not send them" % (l, p, cnt, q))
        # ---- local operators in the level layouts -------------------------------------------------
        for l in range(self.n_dist):
            d = self.dl[l]
            rp, ci, va, src_own = raw[l]["A"]
            A_l = full.levels[l].A
            rg = real_ghosts[l]
            parts_rp, parts_ci, parts_va, d.A_src = [], [], [], []
            for gset in (d.ghost_lo, None, d.ghost_hi):
                if gset is None:
                    parts_rp.append((rp[1:] - rp[:-1]).long())
                    parts_ci.append(ci)
                    parts_va.append(va)
                    d.A_src.append(src_own)
                    continue
                if gset.numel() and rg.numel():
                    pos = torch.searchsorted(rg, gset).clamp(max=rg.numel() - 1)
                    is_real = rg[pos] == gset
                else:
                    is_real = torch.zeros(gset.numel(), dtype=torch.bool, device=gset.device)
                cnt, gci, gva, gsrc = _gather_rows(A_l, gset[is_real], with_source=True)
                full_cnt = torch.zeros(gset.numel(), dtype=torch.long, device=gset.device)
                full_cnt[is_real] = cnt
                parts_rp.append(full_cnt)
                parts_ci.append(gci)
                parts_va.append(gva)
                d.A_src.append(gsrc)
            counts = torch.cat(parts_rp)
            rp_loc = torch.zeros(d.n_tot + 1, dtype=I32, device=va.device)
            rp_loc[1:] = torch.cumsum(counts, 0).to(I32)
            d.A = DeviceCSR(rp_loc, d.to_local(torch.cat(parts_ci)).to(I32).contiguous(),
                            torch.cat(parts_va).contiguous(), (d.n_tot, d.n_tot))
            d.rows_global = torch.cat([d.ghost_lo, torch.arange(d.lo, d.hi, device=va.device), d.ghost_hi])
            d.dinv = self.ops.csr_inverse_diagonal(d.A)
            rp, ci, va, src_r = raw[l]["R"]               # rows: level l+1, columns: level l
            rp_p, ci_p, va_p, src_p = raw[l]["P"]         # rows: level l,   columns: level l+1
            d.R_src = [src_r]
            # P rows: owned rows + the inner ghost layers (real), other ghost rows empty
            P_l = full.levels[l].P
            pg = p_ghost_rows[l]
            pp_rp, pp_ci, pp_va, d.P_src = [], [], [], []
            for gset in (d.ghost_lo, None, d.ghost_hi):
                if gset is None:
                    pp_rp.append((rp_p[1:] - rp_p[:-1]).long())
                    pp_ci.append(ci_p)
                    pp_va.append(va_p)
                    d.P_src.append(src_p)
                    continue
                if gset.numel() and pg.numel():
                    pos = torch.searchsorted(pg, gset).clamp(max=pg.numel() - 1)
                    is_real = pg[pos] == gset
                else:
                    is_real = torch.zeros(gset.numel(), dtype=torch.bool, device=gset.device)
                cnt, gci, gva, gsrc = _gather_rows(P_l, gset[is_real], with_source=True)
                full_cnt = torch.zeros(gset.numel(), dtype=torch.long, device=gset.device)
                full_cnt[is_real] = cnt
                pp_rp.append(full_cnt)
                pp_ci.append(gci)
                pp_va.append(gva)
                d.P_src.append(gsrc)
            rp_ploc = torch.zeros(d.n_tot + 1, dtype=I32, device=va.device)
            rp_ploc[1:] = torch.cumsum(torch.cat(pp_rp), 0).to(I32)
            ci_ploc, va_ploc = torch.cat(pp_ci), torch.cat(pp_va).contiguous()
            if l + 1 < self.n_dist:
                nxt = self.dl[l + 1]
                d.R = DeviceCSR(nxt.embed_rows(rp), d.to_local(ci).to(I32).contiguous(), va.contiguous(),
                                (nxt.n_tot, d.n_tot))
                d.P = DeviceCSR(rp_ploc, nxt.to_local(ci_ploc).to(I32).contiguous(), va_ploc, (d.n_tot, nxt.n_tot))
            else:                                         # next level is replicated on every rank
                nrows = self.bounds[l + 1][self.rank + 1] - self.bounds[l + 1][self.rank]
                d.R = DeviceCSR(rp.contiguous(), d.to_local(ci).to(I32).contiguous(), va.contiguous(),
                                (nrows, d.n_tot))
                d.P = DeviceCSR(rp_ploc, ci_ploc.contiguous(), va_ploc, (d.n_tot, full.levels[l + 1].n))
        if getattr(full, "use_packed", False):
            for l, d in enumerate(self.dl):
                d.A.pack()
                # local blocks are whole grid lines: their transfers are grid transfers with these line lengths
                # (row patterns relative to a column-base map instead of a packed CSR stream)
                wf, wc = math.isqrt(sizes[l]), math.isqrt(sizes[l + 1])
                hint = (wf, wc) if (side is not None and wf * wf == sizes[l] and wc * wc == sizes[l + 1]) else None
                for M in (d.R, d.P):
                    # (DeviceCSR.pack is the same method whatever ops module drives the kernels: always takes the hint)
                    M.pack(line_strides=hint)
        # ---- all-gather plumbing for the first replicated level ----------------------------------
        L = self.n_dist
        cb = self.bounds[L]
        self.ag_max = max(cb[p + 1] - cb[p] for p in range(self.world))
        self.ag_send = torch.zeros(self.ag_max, dtype=F64, device=self.device)
        self.ag_recv = torch.zeros(self.ag_max * self.world, dtype=F64, device=self.device)
        idx = np.concatenate([np.arange(cb[p + 1] - cb[p]) + p * self.ag_max for p in range(self.world)])
        self.ag_index = torch.from_numpy(idx.astype(np.int32)).to(self.device)
        self.ag_rows = cb[self.rank + 1] - cb[self.rank]
        self.use_tail_graph = True
        self._tail_graphs = {}
        self.partials = torch.empty(max(1024, self.ops.partials_count(self.dl[0].n_tot)), dtype=F64,
                                    device=self.device)
        self.norm2 = torch.zeros(1, dtype=F64, device=self.device)

    # ------------------------------------------------------------------------------------------
    @classmethod
    def from_problem(cls, A, transfers, device, ops_mod=None, **kw):
        from .hierarchy import Hierarchy
        return cls(Hierarchy(A, transfers, device, ops_mod=ops_mod), device, ops_mod=ops_mod, **kw)

    @staticmethod
    def _anchors(R):
        """anchor[c] = column (fine row) of the largest entry of row c of R = P^T."""
        n = R.shape[0]
        counts = (R.rowptr[1:] - R.rowptr[:-1]).long()
        rows = torch.repeat_interleave(torch.arange(n, device=R.vals.device), counts)
        # largest value per row, then the first column attaining it
        mx = torch.full((n,), -math.inf, dtype=F64, device=R.vals.device)
        mx = mx.scatter_reduce(0, rows, R.vals, reduce="amax", include_self=True)
        cand = torch.where(R.vals >= mx[rows], R.colidx.long(), torch.full_like(rows, 2 ** 62))
        anchor = torch.full((n,), 2 ** 62, dtype=torch.long, device=R.vals.device)
        anchor = anchor.scatter_reduce(0, rows, cand, reduce="amin", include_self=True)
        anchor = torch.where(anchor == 2 ** 62, torch.zeros_like(anchor), anchor)   # empty rows
        return anchor

    # ---- communication -----------------------------------------------------------------------------
    def exchange(self, d, vec):
        """Fill the ghost segment of `vec` (layout of level d) from the owning ranks."""
        if not d.recv and not d.send:
            return
        self.n_exchanges += 1
        p2p, landing = [], []
        for q, idx, buf in d.send:
            if isinstance(idx, tuple):                      # contiguous run of owned rows: no pack kernel
                buf = vec[idx[0]:idx[1]]
            else:
                self.ops.gather(idx, vec, buf)
            if self.host_staged:
                buf = buf.cpu()
            p2p.append(dist.P2POp(dist.isend, buf, q, group=self.group))
        for q, off, cnt in d.recv:
            dst = vec[off:off + cnt]
            if self.host_staged:
                host = torch.empty(cnt, dtype=F64)
                landing.append((dst, host))
                dst = host
            p2p.append(dist.P2POp(dist.irecv, dst, q, group=self.group))
        for req in dist.batch_isend_irecv(p2p):
            req.wait()
        for dst, host in landing:
            dst.copy_(host)

    def set_rhs(self, rhs):
        d = self.dl[0]
        full = torch.from_numpy(np.ascontiguousarray(np.asarray(rhs, dtype=np.float64).reshape(-1)))
        d.b.copy_(full[d.rows_global.cpu()])          # ghost rows carry real matrix rows: their rhs too

    def set_x(self, x):
        d = self.dl[0]
        full = torch.from_numpy(np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(-1)))
        d.x[d.own].copy_(full[d.lo:d.hi])

    def gather_solution(self):
        """Full fine-level iterate on every rank (host array); test / output helper."""
        d = self.dl[0]
        parts = [None] * self.world
        dist.all_gather_object(parts, d.x[d.own].cpu().numpy(), group=self.group)
        return np.concatenate(parts)

    # ---- the cycle ---------------------------------------------------------------------------------------
    def _smooth(self, d, steps, omega, x_is_zero=False, deep=False, exchanged=False, want_residual=False):
        """`steps` Jacobi sweeps on the whole local block.  deep: the ghosts of x (of b when the
        iterate starts from zero) are exchanged ONCE (not at all when the caller knows they are
        exact: `exchanged`); sweep k is then exact up to ghost layer halo_depth - k, which is all
        the next sweep needs.  Otherwise one exchange per sweep.
        Returns True when d.r = b - A x was produced along the way (fused pass, want_residual)."""
        o = self.ops
        fused = getattr(o, "stencil_smooth_available", None)
        if deep and steps > 0 and fused is not None and fused(d.A):
            # no message between the sweeps: they (and the residual) run as fused passes over the local
            # block -- the same bits as one launch per sweep (lmg_stencil_smooth)
            if not exchanged:
                self.exchange(d, d.b if x_is_zero else d.x)
            left, zero = steps, x_is_zero
            while left > 0:
                k = min(left, o.FUSED_MAX_SWEEPS)
                left -= k
                o.stencil_smooth(d.A, None if zero else d.x, d.b, omega, k, d.tmp,
                                 d.r if (want_residual and left == 0) else None)
                d.x, d.tmp = d.tmp, d.x
                zero = False
            return want_residual
        if x_is_zero and steps > 0:
            if deep and not exchanged:
                self.exchange(d, d.b)
            # first sweep from zeros: x = omega * (D^-1 b), no halo of x needed
            o.vmul(omega, d.dinv, d.b, d.tmp)
            d.x, d.tmp = d.tmp, d.x
            steps -= 1
        elif x_is_zero:
            o.zero(d.x)
        elif deep and not exchanged:
            self.exchange(d, d.x)
        for _ in range(steps):
            if not deep:
                self.exchange(d, d.x)
            o.csr_jacobi(d.A, d.x, d.b, omega, d.tmp)
            d.x, d.tmp = d.tmp, d.x
        return False

    def _owned_gs_schedule(self, d):
        """Level schedule of the exact forward sweep over the OWNED rows of a local block (dependencies among owned rows
        only: ghost columns are constants of the sweep), as local row indices; built once per level."""
        if getattr(d, "gs_sched", None) is None:
            import scipy.sparse as sp
            rp = d.A.rowptr.cpu().numpy().astype(np.int64)
            ci = d.A.colidx.cpu().numpy().astype(np.int64)
            lo, hi = d.n_lo, d.n_lo + d.n_own
            rows = np.repeat(np.arange(d.n_tot, dtype=np.int64), np.diff(rp))
            keep = (rows >= lo) & (rows < hi) & (ci >= lo) & (ci < hi)
            pat = sp.csr_matrix((np.ones(int(keep.sum()), dtype=np.int8), (rows[keep] - lo, ci[keep] - lo)),
                                shape=(d.n_own, d.n_own))
            pat.sort_indices()
            sched = self.ops.build_gs_schedule(pat, "lexicographic", self.device)
            sched.d_rows += lo                                   # local row numbering of the block ([ghosts | owned | ghosts])
            prep = getattr(self.ops, "gs_prepare", None)
            if prep is not None:
                prep(d.A, sched)
            d.gs_sched = sched
        return d.gs_sched

    def _smooth_gs(self, d, steps, x_is_zero):
        """`steps` processor-block Gauss-Seidel sweeps: exchange the ghosts of x, then an exact forward sweep over the
        owned rows with those ghost values (Multigrid.py:88 inside the block)."""
        o = self.ops
        if x_is_zero:
            o.zero(d.x)
        wave = getattr(o, "stencil_gs_available", None)
        # the wavefront kernel relaxes EVERY row of the matrix it is given: only with the classic one-layer halo are the
        # ghost rows of the local operator empty (and therefore left alone)
        use_wave = wave is not None and self.halo_depth == 1 and wave(d.A)
        for k in range(steps):
            if not (x_is_zero and k == 0):
                self.exchange(d, d.x)                           # (a zero iterate has zero ghosts)
            if use_wave:
                o.stencil_gs(d.A, d.x, d.b, 1)
            else:
                o.csr_gs_schedule(d.A, d.x, d.b, self._owned_gs_schedule(d), 1)

    def cycle(self, smoother, steps, omega=1.0, l=0, x_is_zero=False):
        """One V-cycle from level l down.  Returns the number of ghost layers on which this level's
        iterate is exact afterwards (what the caller may prolongate from without a message)."""
        if smoother not in ("Jacobi", "GaussSeidel"):
            raise ValueError("the distributed V-cycle supports the smoothers 'Jacobi' and 'GaussSeidel' "
                             "(processor-block Gauss-Seidel), not %r" % (smoother,))
        gs = smoother == "GaussSeidel"
        o = self.ops
        d = self.dl[l]
        D = self.halo_depth
        # nu sweeps + the residual on the ghost layers the restriction reads consume nu + 1 + r_need
        # layers of one exchange; otherwise (deeper cycles, wide transfers) exchange before every use.
        # Block Gauss-Seidel relaxes owned rows only: the ghost layers are never brought forward, every use exchanges.
        deep = (not gs) and steps >= 1 and steps + 1 + max(1, self.r_need[l]) <= D
        if gs:
            self._smooth_gs(d, steps, x_is_zero)
            have_r = False
        else:
            have_r = self._smooth(d, steps, omega, x_is_zero, deep, want_residual=True)
        if not deep:
            self.exchange(d, d.x)
        if not have_r:
            o.csr_residual_norm2(d.A, d.x, d.b, d.r, None, None)
        if not deep:
            self.exchange(d, d.r)
        # The prolongation is applied on the ghost layers too (real P rows there).  The iterate is
        # exact on D - steps layers before the correction; if the correction is exact on V >= steps
        # of them, post-smoothing needs no message and leaves V - steps exact layers behind.
        want_local = deep and 2 * steps <= D
        V = 0
        if l + 1 < self.n_dist:
            nxt = self.dl[l + 1]
            o.csr_spmv(d.R, d.r, nxt.b, 1.0, 0.0)
            v_next = self.cycle(smoother, steps, omega, l + 1, x_is_zero=True)
            if want_local:
                # largest V whose P rows only read coarse ghosts that are already exact
                V = max([k for k in range(steps, D - steps + 1) if self.p_need[l][k] <= v_next], default=0)
            if V == 0:
                self.exchange(nxt, nxt.x)
                V = D - steps if want_local else 0
            o.csr_spmv(d.P, nxt.x, d.x, 1.0, 1.0)
        else:
            fl = self.full.levels[l + 1]
            o.csr_spmv(d.R, d.r, self.ag_send[:self.ag_rows], 1.0, 0.0)
            if self.host_staged:
                recv = torch.empty(self.ag_recv.numel(), dtype=F64)
                dist.all_gather_into_tensor(recv, self.ag_send.cpu(), group=self.group)
                self.ag_recv.copy_(recv)
            else:
                dist.all_gather_into_tensor(self.ag_recv, self.ag_send, group=self.group)
            o.gather(self.ag_index, self.ag_recv, fl.b)
            self._replicated_tail(smoother, steps, omega, l + 1)
            o.csr_spmv(d.P, fl.x, d.x, 1.0, 1.0)      # the replicated correction is complete on every rank
            V = D - steps if want_local else 0
        local_up = want_local and V >= steps
        if gs:
            self._smooth_gs(d, steps, False)
            return 0
        self._smooth(d, steps, omega, False, deep, exchanged=local_up)
        return V - steps if local_up else 0

    def _replicated_tail(self, smoother, steps, omega, l):
        """The part of the cycle below the distributed levels: purely local work on the
        replicated hierarchy, so it is captured once into a hipGraph and replayed (the RCCL
        calls above it stay eager)."""
        def run():
            if l + 1 == len(self.full.levels):
                self.full.coarse_solve()
            else:
                self.full.cycle(smoother, steps, omega, l=l, x_is_zero=True)

        if not self.use_tail_graph or not hasattr(self.ops, "CapturedGraph") or self.device.type != "cuda":
            run()
            return
        key = (smoother, steps, omega, l)
        g = self._tail_graphs.get(key)
        if g is None:
            prep = getattr(self.full, "prepare_smoother", None)
            if prep is not None:
                prep(smoother, "lexicographic", l)               # (device -> host reads, allocations: not inside a capture)
            before = [(lev.x, lev.tmp) for lev in self.full.levels]
            g = self.ops.CapturedGraph()
            with g:
                run()
            after = [(lev.x, lev.tmp) for lev in self.full.levels]
            if any(a[0] is not b[0] for a, b in zip(before, after)):
                raise RuntimeError("ping-pong buffers did not return to their slots")
            self._tail_graphs[key] = g
        g.launch()

    def rebuild_numeric(self, new_vals):
        """Galerkin rebuild after the VALUES of the fine matrix changed (config #5).  The numeric
        SpGEMM passes run on the replicated hierarchy of every rank (no communication); the local
        operators then take their values again from where they were cut, and their lossless twins
        are refreshed."""
        self.full.rebuild_numeric(new_vals)
        for l, d in enumerate(self.dl):
            lev = self.full.levels[l]
            for local, replicated, src in ((d.A, lev.A, d.A_src), (d.P, lev.P, d.P_src), (d.R, lev.R, d.R_src)):
                if local.nnz:
                    local.vals.copy_(_take(replicated.vals, src))
                    local.repack_values()
            d.dinv = self.ops.csr_inverse_diagonal(d.A)
        self._tail_graphs = {}           # the replicated tail's graph holds pointers of the old twins

    def residual_norm(self):
        """||b - A x||_2 over all ranks: local fused sum of squares + all-reduce of 8 bytes."""
        d = self.dl[0]
        self.exchange(d, d.x)
        self.ops.csr_residual_norm2(d.A, d.x, d.b, d.r, None, None)
        r_own = d.r[d.own]                              # ghost rows are real rows: count the owned ones only
        self.ops.dot(r_own, r_own, self.partials, self.norm2)
        if self.host_staged:
            h = self.norm2.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            return math.sqrt(h.item())
        dist.all_reduce(self.norm2, op=dist.ReduceOp.SUM, group=self.group)
        return math.sqrt(self.norm2.item())

    def make_step(self, smoother, steps, omega, graph=False):
        """One V-cycle as a callable.  `graph` is accepted for symmetry with Hierarchy.captured_cycle: the replicated tail
        below the distributed levels always replays from a hipGraph (`use_tail_graph`); the kernel segments between two
        messages of the distributed levels are launched eagerly -- capturing each of them was measured at world size 1
        (2049^2: 0.592 ms with segment graphs vs 0.576 eager) and dropped, RCCL point-to-point inside a captured graph has
        not been tried on this pool."""
        def step():
            self.cycle(smoother, steps, omega)
        return step

    # what bench.py reports the roofline on
    @property
    def local_hierarchy(self):
        return self.full

    @property
    def fine_local_matrix(self):
        return self.dl[0].A

    @property
    def fine_local_rows(self):
        return self.dl[0].n_own
