"""Host-side model of the launch-order packer (`pack_order_kernel`, csrc/wbcqp_device.hpp) -- same integer arithmetic, one
Python loop per device step.  Used by tests (the device result is compared with it class by class) and by tools that predict
a launch's makespan; the product never calls it.

Problem: resident workgroups take QPs from a queue in launch order (list scheduling); a QP costs about SETUP_ITERS + iterations
units.  Longest-first order with ~4 QPs per workgroup ends up to one short QP above the mean.  Here the longest-first order is
dealt out boustrophedon to SUBS sub-problems, each packed into its share of the workgroups for a capacity C: a bin takes the
largest QP left, then QPs whose size is nearest (room / number of QPs the room still holds at the mean size left), the last
two chosen to fill the room exactly when such a pair exists.  The smallest of TRIALS capacities that fits wins; the QPs are
emitted by start time, sub-problems interleaved -- list scheduling reproduces the packing, or improves on it.
"""
import numpy as np

SETUP_ITERS = 7      # kSetupIters: setup cost of a Talos QP in units of one active-set iteration (115.8 k vs 14.7 k cycles)
SUBS = 32            # kPackSubs
TRIALS = 4           # kPackTrials
MAX_ITEMS = 128      # kPackMaxItems
MAX_CLASS = 63


def _hi(mask):
    return mask.bit_length() - 1 if mask else -1


def _lo(mask):
    return (mask & -mask).bit_length() - 1 if mask else -1


def _below(k):  # bits 0..k
    return (1 << (k + 1)) - 1 if k >= 0 else 0


def _nearest(mask, t2):
    """set bit k of mask nearest to t2/2, the larger one on a tie; -1 if none"""
    fl = t2 >> 1 if t2 >= 0 else -1
    kl = _hi(mask & _below(min(fl, MAX_CLASS)))
    kh = _lo(mask & ~_below(min(fl, MAX_CLASS)))
    if kl < 0:
        return kh
    if kh < 0:
        return kl
    return kh if (2 * kh - t2) <= (t2 - 2 * kl) else kl


def pack_group(classes, nbin, cap, A=SETUP_ITERS):
    """classes: descending array of one XCD group's classes.  Returns [(start, local index)] in selection order or None."""
    n = len(classes)
    cnt = [0] * (MAX_CLASS + 1)
    for k in classes:
        cnt[int(k)] += 1
    gend = [0] * (MAX_CLASS + 1)
    acc = 0
    for k in range(MAX_CLASS, -1, -1):
        acc += cnt[k]
        gend[k] = acc
    avail = sum(1 << k for k in range(MAX_CLASS + 1) if cnt[k])
    tot = int(sum(int(k) for k in classes)) + A * n
    events = []

    def take(k, start):
        nonlocal avail, n, tot
        events.append((start, gend[k] - cnt[k]))
        cnt[k] -= 1
        if cnt[k] == 0:
            avail &= ~(1 << k)
        n -= 1
        tot -= k + A

    for _ in range(nbin):
        room = cap
        first = True
        while n > 0:
            kmin = _lo(avail)
            if room < kmin + A:
                break
            kmax = min(MAX_CLASS, room - A)
            if first:
                k = _hi(avail & _below(kmax))
                first = False
            else:
                j = 1
                while j < 8 and (2 * j + 1) * tot <= 2 * room * n:
                    j += 1
                while j > 1 and j * (kmin + A) > room:
                    j -= 1
                if j == 1:
                    k = _hi(avail & _below(kmax))
                else:
                    lim = min(kmax, room - (j - 1) * (kmin + A) - A)
                    t2 = (2 * room) // j - 2 * A
                    k = _nearest(avail & _below(lim), t2)
                    if j == 2:
                        need = room - 2 * A
                        cand = 0
                        for k2 in range(max(0, need - MAX_CLASS), min(MAX_CLASS, need) + 1):
                            k3 = need - k2
                            if cnt[k2] and cnt[k3] and (k2 != k3 or cnt[k2] >= 2):
                                cand |= 1 << k2
                        bp = _nearest(cand, t2)
                        if bp >= 0:
                            k = max(bp, need - bp)
            if k < 0:
                break
            take(k, cap - room)
            room -= k + A
    return events if n == 0 else None


def packs(total, resident, n_groups=1):
    """the host's condition for launching pack_order_kernel (wbcqp_api.hip, launch())"""
    return (n_groups == 1 and resident % SUBS == 0 and total % SUBS == 0 and resident < total <= 8 * resident
            and total // SUBS <= MAX_ITEMS)


def pack_order(order, iters, resident=256, A=SETUP_ITERS):
    """order: longest-first permutation (what schedule_kernel leaves); iters: previous iteration counts.  Returns the packed
    order: slot SUBS * t + w holds the t-th QP (by start time) of sub-problem w; a sub-problem that fits no trial keeps the
    longest-first order of its share."""
    order = np.asarray(order)
    total = len(order)
    assert packs(total, resident)
    out = np.empty_like(order)
    per = resident // SUBS
    for w in range(SUBS):
        ranks = [SUBS * e + (SUBS - 1 - w if e & 1 else w) for e in range(total // SUBS)]
        jobs = order[ranks]
        cls = np.clip(np.asarray(iters)[jobs], 0, MAX_CLASS).astype(int)
        assert np.all(np.diff(cls) <= 0), "order must be longest-first"
        lb = -(-(int(cls.sum()) + A * len(cls)) // per)
        c0 = max(lb, int(cls.max()) + A)
        ev = None
        for t in range(TRIALS):
            ev = pack_group(cls, per, c0 + t, A)
            if ev is not None:
                break
        if ev is None:
            out[w::SUBS] = jobs
        else:
            rank = sorted(range(len(ev)), key=lambda e: (ev[e][0], e))
            out[w::SUBS] = [jobs[ev[e][1]] for e in rank]
    return out


def makespan(costs_in_order, workers=256):
    """list scheduling: the next QP of the order goes to the first worker that is free"""
    import heapq
    h = [0.0] * workers
    heapq.heapify(h)
    for c in costs_in_order:
        heapq.heappush(h, heapq.heappop(h) + float(c))
    return max(h)


def makespan_hw(costs_in_order, per_engine=8, engines=32):
    """the hardware's own dispatch, as measured (tools/ubench/dispatch_order.hip): workgroup i belongs to XCD i % 8 and to
    shader engine (i / 8) % 4 of it, goes to the first free CU of that engine, and nothing behind it in the XCD overtakes it"""
    import heapq
    heaps = [[0.0] * per_engine for _ in range(engines)]
    tx = [0.0] * 8
    end = 0.0
    for i, c in enumerate(costs_in_order):
        hq = heaps[i % engines]
        free = heapq.heappop(hq)
        tx[i % 8] = max(tx[i % 8], free)
        heapq.heappush(hq, tx[i % 8] + float(c))
        end = max(end, tx[i % 8] + float(c))
    return end
