"""Work split of the pair-sharded form on scripted attempts (no GPU): attempts and fused batches per rank for the first registration of a
serpentine (cold: history-driven speculation, blind chunk starts, chunks by pair count) and for the next one (path memory: predicted chunk
starts, chunks by predicted attempts, plan read off the prediction) -- lock-step threads with an in-process all-gather.
    python tools/work_split.py > profiles/r04_work_split.txt"""
import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, threading
from scripted import ScriptedAttemptEngine
from imagestitch_amd.grid import GridRegistrar, serpentine_directions
SHAPE=(2048,2048)
def clean(rows, cols):
    acc=[]
    for c in range(cols):
        d_col = 1 if c % 2 == 0 else 3
        for r in range(rows-1): acc.append({(d_col,i):(3,4) for i in range(1,4)})
        if c<cols-1: acc.append({(2,i):(3,4) for i in range(1,4)})
    return acc
def run(rows, cols, world, steps=2):
    acc=clean(rows,cols); P=len(acc); shapes=[SHAPE]*(P+1)
    bar=threading.Barrier(world); slots=[None]*world; out={}
    def work(rank):
        reg=GridRegistrar(ScriptedAttemptEngine(SHAPE,0.2,acc), roiRatio=0.2, directIncre=1, window=48); reg.native=False
        def ag(p):
            slots[rank]=np.asarray(p,np.int32); bar.wait(); g=np.stack(slots); bar.wait(); return g
        per=[]
        for s in range(steps):
            a0,b0=reg.stats['attempts'],reg.stats['batches']
            reg.register_sharded(list(range(P+1)), shapes, 1, rank, world, ag)
            per.append((reg.stats['attempts']-a0, reg.stats['batches']-b0))
        out[rank]=per
    ts=[threading.Thread(target=work,args=(r,)) for r in range(world)]
    [t.start() for t in ts]; [t.join() for t in ts]
    return [out[r] for r in range(world)]
for rows,cols in ((10,9),(32,32)):
    for world in (1,2,4,8):
        res=run(rows,cols,world)
        cold=[r[0] for r in res]; hot=[r[1] for r in res]
        print(rows,cols,'N=%d'%world,'cold attempts',[c[0] for c in cold],'batches',[c[1] for c in cold],'| hot attempts',[h[0] for h in hot],'batches',[h[1] for h in hot])
