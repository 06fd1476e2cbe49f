"""Config 4, phase 1: what would another ORDER of the same trees buy?  Event simulation on the recorded tree sizes of one
steady-state iteration (gpurun_out/c4_pred.npz from tools/c4_predict.py: 65 536 trees, mean 323 leaves), 16 384 resident lane
groups, one leaf-time per leaf, trees cut at 511 leaves (the finisher has the rest).
  (a) the shipped schedule: one queue, a group takes the next tree when it has ended one;
  (b) trees parked at further doubling boundaries and re-queued IN THE SAME LAUNCH, groups preferring the shallowest ready
      level (bfs: every tree gets its short first part early, the launch ends on the uniform 256-leaf pieces of the last
      level) or the deepest (dfs); 3 leaf-times charged per take-up;
  (c) the same levels as separate launches.
    python tools/sim_c4_schedule.py   ->  profiles/r05_c4_schedule_sim.txt"""
import numpy as np, heapq, sys
d=np.load('gpurun_out/c4_pred.npz'); nl=d['nleap'].astype(np.int64)
N=len(nl); CAP=511
hist=np.bincount(np.minimum(nl,2047))
print("mean",nl.mean(),"share >511:",(nl>511).mean(),">255:",(nl>255).mean(),">127:",(nl>127).mean(),">63:",(nl>63).mean())
main=np.minimum(nl,CAP)
print("main leaves",main.sum(),"fin leaves",(nl-main).sum(), "fin trees", (nl>CAP).sum())
G=16384
# (a) current: greedy queue over G groups (ignoring wave lock-step and step_align)
def greedy(jobs, G):
    h=[0]*G; heapq.heapify(h); busy=0
    for L in jobs:
        t=heapq.heappop(h); heapq.heappush(h,t+L); busy+=L
    return max(h), busy
mk,busy=greedy(main.tolist(),G)
print("(a) single queue: makespan",mk,"ideal",busy/G,"eff",busy/G/mk)
# (b) breadth-first multi-level, levels at boundaries B (in leaves): job pieces
def pieces(L,bounds):
    out=[];prev=0
    for b in bounds:
        if L<=prev: break
        out.append(min(L,b)-prev); prev=b
    return out
def multilevel(bounds, policy, ovh=0):
    # event simulation: groups take jobs; priority by policy among ready queues
    import collections
    nlev=len(bounds)
    queues=[collections.deque() for _ in range(nlev)]
    for i in range(N): queues[0].append(i)
    pcs=[pieces(int(L),bounds) for L in main]
    ev=[(0,g,-1,-1) for g in range(G)]  # (time, group, tree, level) completion events
    heapq.heapify(ev); busy=0; tmax=0; idle=[]
    waiting=[] # idle groups
    pending=N  # trees not finished
    while ev:
        t,g,tr,lv=heapq.heappop(ev)
        if tr>=0:
            if lv+1<len(pcs[tr]): queues[lv+1].append(tr)
            else: pending-=1
            tmax=max(tmax,t)
        waiting.append(g)
        # peek: process all events at same time lazily -- assign work to all waiting groups
        while waiting:
            order=range(nlev) if policy=='bfs' else range(nlev-1,-1,-1)
            q=None
            for l in order:
                if queues[l]: q=l;break
            if q is None: break
            g2=waiting.pop(); tr2=queues[q].popleft(); L=pcs[tr2][q]+(ovh if q>0 else 0)
            busy+=L; heapq.heappush(ev,(t+L,g2,tr2,q))
        if pending==0: break
    return tmax,busy
for bounds in ([511],[63,511],[63,255,511],[127,255,511],[63,127,255,511],[255,511],[31,63,127,255,511]):
    for pol in ('bfs','dfs'):
        mk,b=multilevel(bounds,pol,ovh=3)
        print(bounds,pol,"makespan",mk,"eff",main.sum()/G/mk)
print("--- separate launches (barrier between phases), greedy queue in each")
for bounds in ([63,511],[255,511],[63,255,511],[127,255,511],[31,127,255,511],[63,127,255,511]):
    tot=0; prev=0; parts=[]
    for b in bounds:
        jobs=(np.minimum(main,b)-prev); jobs=jobs[main>prev]
        mk,_=greedy((jobs+ (3 if prev>0 else 0)).tolist(),G); tot+=mk; parts.append((len(jobs),mk)); prev=b
    print(bounds,"total",tot,parts,"eff",main.sum()/G/tot)
