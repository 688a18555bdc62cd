"""Cost model of march_kernel's scheduling: how many wave-level VALU instructions a ray costs under a scheduling
policy, given the measured per-phase instruction counts and the measured transition rates of a configuration.
Planning tool only (no GPU, no oracle): rays are a Markov chain MARCH -> {MARCH, HIT, ENDED}, HIT -> {MARCH, ENDED}
with the rates of tools/diag_march.py; a workgroup's waves advance in parallel, each paying the wave-level cost of the
body it executes whatever the number of active lanes.

    python tools/sim_pool.py [c3|c5]
"""
import random
import sys

CFG = {
    # p_hit / p_end: per march iteration; p_break: per hit           (profiles/r03_diag_c3.txt, r03_diag_c5.txt)
    "c3": dict(p_hit=0.356, p_end=0.166, p_break=0.20),
    "c5": dict(p_hit=0.167, p_end=0.0053, p_break=0.07),
}
# wave-level VALU instructions per execution of a body (static counts of the shipped kernel's ISA, hot path)
COST = dict(march=310, hit=350, end=100, refill=150, pass_=40, xchg=30)

M, H, E, I = 0, 1, 2, 3


class Ray:
    __slots__ = ("state",)

    def __init__(self):
        self.state = M


def step_march(ray, c, rng):
    u = rng.random()
    if u < c["p_hit"]:
        ray.state = H
    elif u < c["p_hit"] + c["p_end"]:
        ray.state = E


def step_hit(ray, c, rng):
    ray.state = E if rng.random() < c["p_break"] else M


def baseline(c, rng, n_rays=200000, t_hit=24, t_end=32, max_iters=3):
    """the shipped policy: one ray per lane, thresholds"""
    lanes = [None] * 64
    done = 0
    cost = 0
    lane_sum = {"march": 0, "hit": 0, "end": 0, "refill": 0}
    execs = {"march": 0, "hit": 0, "end": 0, "refill": 0}
    issued = 0
    while done < n_rays:
        cost += COST["pass_"]
        idle = [i for i in range(64) if lanes[i] is None]
        if idle and issued < n_rays:
            cost += COST["refill"]
            execs["refill"] += 1
            lane_sum["refill"] += len(idle)
            for i in idle:
                lanes[i] = Ray()
                issued += 1
        iters = 0
        while True:
            nm = sum(1 for r in lanes if r and r.state == M)
            nh = sum(1 for r in lanes if r and r.state == H)
            ne = sum(1 for r in lanes if r and r.state == E)
            if nm == 0 or nh >= t_hit or ne >= t_end:
                break
            if iters >= max_iters and nh + ne > 0:
                break
            iters += 1
            cost += COST["march"]
            execs["march"] += 1
            lane_sum["march"] += nm
            for r in lanes:
                if r and r.state == M:
                    step_march(r, c, rng)
        nm = sum(1 for r in lanes if r and r.state == M)
        nh = sum(1 for r in lanes if r and r.state == H)
        capped = iters >= max_iters
        if nh and (nm == 0 or capped or nh >= t_hit):
            cost += COST["hit"]
            execs["hit"] += 1
            lane_sum["hit"] += nh
            for r in lanes:
                if r and r.state == H:
                    step_hit(r, c, rng)
        nm = sum(1 for r in lanes if r and r.state == M)
        ne = sum(1 for r in lanes if r and r.state == E)
        if ne and (nm == 0 or capped or ne >= t_end):
            cost += COST["end"]
            execs["end"] += 1
            lane_sum["end"] += ne
            for i in range(64):
                if lanes[i] and lanes[i].state == E:
                    lanes[i] = None
                    done += 1
    return cost / done, {k: lane_sum[k] / max(1, execs[k]) for k in execs}


def private_pool(c, rng, parked=48, n_rays=200000, t_hit=56, t_end=56, swap_min=8):
    """wave-private pool: 64 lanes + `parked` slots in LDS, no cross-wave traffic.  Before a body runs, lanes whose ray
    is in another state swap with parked rays in the wanted state."""
    lanes = [None] * 64
    park = []
    done = issued = 0
    cost = 0
    lane_sum = {"march": 0, "hit": 0, "end": 0, "refill": 0}
    execs = {"march": 0, "hit": 0, "end": 0, "refill": 0, "xchg": 0}
    moved = 0

    def count(st):
        return sum(1 for r in lanes if r and r.state == st), sum(1 for r in park if r.state == st)

    def gather(st):
        """bring parked rays of state st into lanes that hold something else (or nothing)"""
        nonlocal cost, moved
        src = [r for r in park if r.state == st]
        if not src:
            return
        # empty lanes first, then lanes in other states while there is room to park them
        targets = [i for i in range(64) if lanes[i] is None] + [i for i in range(64) if lanes[i] and lanes[i].state != st]
        n = 0
        for i in targets:
            if not src:
                break
            r = src.pop()
            park.remove(r)
            if lanes[i] is not None:
                park.append(lanes[i])
            lanes[i] = r
            n += 1
        if n:
            cost += COST["xchg"]
            execs["xchg"] += 1
            moved += n

    while done < n_rays:
        cost += COST["pass_"]
        lm, pm = count(M)
        lh, ph = count(H)
        le, pe = count(E)
        nidle = sum(1 for r in lanes if r is None)
        total = 64 - nidle + len(park)
        # choose the body with the most rays available, preferring the expensive waits
        if lh + ph >= min(t_hit, total) and lh + ph > 0:
            if ph >= swap_min or lh < 64:
                gather(H)
            n = sum(1 for r in lanes if r and r.state == H)
            cost += COST["hit"]
            execs["hit"] += 1
            lane_sum["hit"] += n
            for r in lanes:
                if r and r.state == H:
                    step_hit(r, c, rng)
            continue
        if le + pe >= min(t_end, total) and le + pe > 0:
            gather(E)
            n = sum(1 for r in lanes if r and r.state == E)
            cost += COST["end"]
            execs["end"] += 1
            lane_sum["end"] += n
            for i in range(64):
                if lanes[i] and lanes[i].state == E:
                    lanes[i] = None
                    done += 1
            idle = [i for i in range(64) if lanes[i] is None]
            if idle and issued < n_rays + 64:
                cost += COST["refill"]
                execs["refill"] += 1
                lane_sum["refill"] += len(idle)
                for i in idle:
                    lanes[i] = Ray()
                    issued += 1
            continue
        # march: park waiting lanes if there is room and marching rays to bring in
        if pm > 0 or nidle > 0:
            # make room: park H/E lanes (each park of a lane ray needs a free slot)
            free = parked - len(park)
            waiting = [i for i in range(64) if lanes[i] and lanes[i].state != M]
            n = 0
            src = [r for r in park if r.state == M]
            for i in waiting:
                if not src:
                    break
                r = src.pop()
                park.remove(r)
                park.append(lanes[i])
                lanes[i] = r
                n += 1
            for i in range(64):
                if lanes[i] is None and src:
                    r = src.pop()
                    park.remove(r)
                    lanes[i] = r
                    n += 1
            if n:
                cost += COST["xchg"]
                execs["xchg"] += 1
                moved += n
        # still idle lanes or lanes waiting with free parking: take fresh rays into them
        free = parked - len(park)
        waiting = [i for i in range(64) if lanes[i] and lanes[i].state != M]
        evict = waiting[:free]
        if evict:
            for i in evict:
                park.append(lanes[i])
                lanes[i] = None
            cost += COST["xchg"]
            execs["xchg"] += 1
            moved += len(evict)
        idle = [i for i in range(64) if lanes[i] is None]
        if idle and issued < n_rays + 64:
            cost += COST["refill"]
            execs["refill"] += 1
            lane_sum["refill"] += len(idle)
            for i in idle:
                lanes[i] = Ray()
                issued += 1
        nm = sum(1 for r in lanes if r and r.state == M)
        if nm == 0:
            # nothing marches: run whatever waits
            lh, ph = count(H)
            if lh + ph:
                gather(H)
                n = sum(1 for r in lanes if r and r.state == H)
                cost += COST["hit"]
                execs["hit"] += 1
                lane_sum["hit"] += n
                for r in lanes:
                    if r and r.state == H:
                        step_hit(r, c, rng)
            else:
                gather(E)
                n = sum(1 for r in lanes if r and r.state == E)
                if n == 0:
                    break
                cost += COST["end"]
                execs["end"] += 1
                lane_sum["end"] += n
                for i in range(64):
                    if lanes[i] and lanes[i].state == E:
                        lanes[i] = None
                        done += 1
            continue
        cost += COST["march"]
        execs["march"] += 1
        lane_sum["march"] += nm
        for r in lanes:
            if r and r.state == M:
                step_march(r, c, rng)
    out = {k: lane_sum[k] / max(1, execs[k]) for k in lane_sum}
    out["xchg/ray"] = execs["xchg"] / done
    out["moved/ray"] = moved / done
    return cost / done, out


if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "c3"
    c = CFG[name]
    rng = random.Random(1)
    pol = dict(c3=(24, 32, 3), c5=(24, 24, 5))[name]
    b, lanes = baseline(c, rng, t_hit=pol[0], t_end=pol[1], max_iters=pol[2])
    print("%s baseline: %.1f wave-VALU per ray; lanes %s" % (name, b, {k: round(v, 1) for k, v in lanes.items()}))
    for parked in (16, 32, 48, 64, 96):
        for th in (40, 48, 56, 64):
            v, lanes = private_pool(c, rng, parked=parked, t_hit=th, t_end=th)
            print("private pool %3d parked, threshold %2d: %.1f (%.2fx)  %s" % (
                parked, th, v, b / v, {k: round(x, 2) for k, x in lanes.items()}))
