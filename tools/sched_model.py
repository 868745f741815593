# Monte-Carlo model of one wave's scheduling: lanes in lockstep; a "job" = [shadow ray] + closest ray (steps each);
# shade call costs SHADE step-equivalents regardless of how many lanes take part.
import random, math, sys
random.seed(1)
SHADE = float(sys.argv[1]) if len(sys.argv) > 1 else 12.0
def ray_steps(mean, sigma):
    mu = math.log(mean) - sigma * sigma / 2
    return max(1, int(random.lognormvariate(mu, sigma) + 0.5))
def new_job(camera):
    s = 0
    if not camera and random.random() < 0.72: s += ray_steps(14, 0.8)
    s += ray_steps(20, 0.7)
    return s
P_CONT = 0.67
def next_job():
    # after shading a vertex: path continues (job with maybe shadow), or ends -> regen camera job (infinite sample queue)
    if random.random() < P_CONT: return new_job(False)
    return new_job(True)

def sim_current(R, total_jobs=200000):
    rem = [new_job(True) for _ in range(64)]
    done = [False] * 64
    t = 0.0; jobs = 0; busy = 0; steps = 0; calls = 0; shaded = 0
    while jobs < total_jobs:
        # trace until R lanes finished (counting those idle at entry? no: finished since entry; all idle lanes were re-launched)
        fin = 0
        while True:
            act = 0
            for i in range(64):
                if rem[i] > 0:
                    rem[i] -= 1; act += 1
                    if rem[i] == 0: fin += 1
            steps += 1; busy += act; t += 1
            if fin >= R or act == fin and all(r == 0 for r in rem): break
        n = 0
        for i in range(64):
            if rem[i] == 0: rem[i] = next_job(); n += 1
        jobs += n; calls += 1; shaded += n; t += SHADE
    return dict(units_per_job=t / jobs, trace_util=busy / (64.0 * steps), shade_lanes=shaded / calls / 64.0, steps_per_call=steps / calls)

def sim_double(T, total_jobs=200000, both=False):
    # 2 slots per lane. slot state: job steps remaining (>0: waiting/in trace), 0 = ready to shade
    slots = [[new_job(True), new_job(True)] for _ in range(64)]
    cur = [0] * 64  # which slot the lane is tracing (or -1)
    t = 0.0; jobs = 0; busy = 0; steps = 0; calls = 0; shaded = 0
    while jobs < total_jobs:
        while True:
            act = 0
            for i in range(64):
                c = cur[i]
                if c >= 0 and slots[i][c] > 0:
                    slots[i][c] -= 1; act += 1
                    if slots[i][c] == 0:
                        o = 1 - c
                        cur[i] = o if slots[i][o] > 0 else -1
                elif c >= 0:
                    o = 1 - c
                    cur[i] = o if slots[i][o] > 0 else -1
            steps += 1; busy += act; t += 1
            ready_lanes = sum(1 for i in range(64) if slots[i][0] == 0 or slots[i][1] == 0)
            idle_lanes = sum(1 for i in range(64) if cur[i] < 0)
            if ready_lanes >= T or idle_lanes >= IDLE_MAX: break
        n = 0
        for i in range(64):
            for c in (0, 1):
                if slots[i][c] == 0:
                    slots[i][c] = next_job(); n += 1
                    if cur[i] < 0: cur[i] = c
                    if not both: break
        jobs += n; calls += 1; shaded += n; t += SHADE * (1.0 if not both else 1.6)
    return dict(units_per_job=t / jobs, trace_util=busy / (64.0 * steps), shade_lanes=shaded / calls / 64.0, steps_per_call=steps / calls)

def sim_pool(P, K, total_jobs=200000, refill_cost=0.08, MINREADY=64):
    # shared pool: rayq FIFO of job lengths, ready count; lanes hold remaining steps
    from collections import deque
    rayq = deque(new_job(True) for _ in range(P))
    rem = [0] * 64
    ready = 0
    t = 0.0; jobs = 0; busy = 0; steps = 0; calls = 0; shaded = 0
    while jobs < total_jobs:
        while True:
            idle = [i for i in range(64) if rem[i] == 0]
            if rayq and (len(idle) >= K):
                for i in idle:
                    if not rayq: break
                    rem[i] = rayq.popleft() + 1   # one idle step while the ray loads
                t += refill_cost
            act = 0
            for i in range(64):
                if rem[i] > 0:
                    rem[i] -= 1; act += 1
                    if rem[i] == 0: ready += 1
            steps += 1; busy += act; t += 1
            if ready >= MINREADY or (not rayq and (act == 0 or ready >= MINREADY2)): break
        n = min(ready, 64)
        for _ in range(n): rayq.append(next_job())
        ready -= n
        jobs += n; calls += 1; shaded += n; t += SHADE
    return dict(units_per_job=t / jobs, trace_util=busy / (64.0 * steps), shade_lanes=shaded / calls / 64.0, steps_per_call=steps / calls)

IDLE_MAX = 65; MINREADY2 = 32
f = lambda d: '  '.join('%s %.3f' % kv for kv in d.items())
for R in (32, 40, 48): print('current R=%d' % R, f(sim_current(R)))
for T in (40, 48, 56, 60):
    for IDLE_MAX in (16, 24, 65): print('double T=%d idlemax=%d' % (T, IDLE_MAX), f(sim_double(T)))
for P in (96, 128, 160, 192):
    for K in (4, 8, 16): print('pool P=%d K=%d' % (P, K), f(sim_pool(P, K)))

def sim_ctx(NCTX, T, IMAX, total_jobs=200000):
    # NCTX contexts per lane; ctx value: >0 steps remaining of pending/active job, 0 = done (awaiting shade)
    ctx = [[new_job(True) for _ in range(NCTX)] for _ in range(64)]
    cur = [0] * 64
    t = 0.0; jobs = 0; busy = 0; steps = 0; calls = 0; shaded = 0
    while jobs < total_jobs:
        while True:
            act = 0
            for i in range(64):
                c = cur[i]
                if c >= 0:
                    ctx[i][c] -= 1; act += 1
                    if ctx[i][c] == 0:
                        cur[i] = -1
                        for o in range(NCTX):
                            if ctx[i][o] > 0: cur[i] = o; break
            steps += 1; busy += act; t += 1
            S = sum(1 for i in range(64) if any(v == 0 for v in ctx[i]))
            SI = sum(1 for i in range(64) if cur[i] < 0)
            if S >= T or SI >= IMAX or act == 0: break
        n = 0
        for i in range(64):
            for c in range(NCTX):
                if ctx[i][c] == 0:
                    ctx[i][c] = next_job(); n += 1
                    if cur[i] < 0: cur[i] = c
                    break
        jobs += n; calls += 1; shaded += n; t += SHADE
    return dict(units_per_job=t / jobs, trace_util=busy / (64.0 * steps), shade_lanes=shaded / calls / 64.0, steps_per_call=steps / calls)
print()
for N in (2, 3):
    for T in (40, 48, 56, 60):
        for I in (8, 16, 24): print('ctx N=%d T=%d I=%d' % (N, T, I), f(sim_ctx(N, T, I)))
