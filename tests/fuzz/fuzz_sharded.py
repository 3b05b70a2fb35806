#!/usr/bin/env python3
"""Randomised runs of the sharded prover against the single-process one: random world size (2 / 4 / 8 dividing the cosets), trace size,
blow-up, FRI steps (incl. groups that straddle cosets), AIR (one or two witness stages), hashCommits, sharded or replicated constant
tree -- every rank of tests/workers/sharded_prove_worker.py asserts that the proof it receives equals the ordinary proof, field by field.
  python tests/fuzz/fuzz_sharded.py SECONDS [oracle|gpu]     oracle: CPU checker backend over gloo (runs anywhere); gpu: the HIP library, every
  rank on cuda:0 exchanging through HIP-IPC windows (worlds of 2 and 4 only: the GPU box allows six processes on its card)"""
import os, sys, subprocess, random, socket, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
BACKEND = sys.argv[2] if len(sys.argv) > 2 else "oracle"
W=os.path.join(ROOT,"tests","workers","sharded_prove_worker.py")
def port():
    s=socket.socket(); s.bind(("127.0.0.1",0)); p=s.getsockname()[1]; s.close(); return p
rnd=random.Random(11)
t0=time.time(); n=0; bad=0
while time.time()-t0 < float(sys.argv[1]):
    eb=rnd.choice([1,2,3,3]); world=rnd.choice([w for w in ((2,4,8) if BACKEND == "oracle" else (2,4)) if w <= (1<<eb)])
    nb=rnd.randint(3,9 if BACKEND == "oracle" else 12); nbe=nb+eb
    steps=[nbe]
    while steps[-1]>3 and len(steps)<5:
        nxt=steps[-1]-rnd.randint(1,5)
        if nxt<1: break
        steps.append(nxt)
    air=rnd.choice(["fib","fib","perm","permref","permres"]); hc=rnd.choice([0,0,1]); ss=rnd.choice([0,1]); pairs=rnd.randint(1,5); im=rnd.choice([0,0,1]); bd=rnd.choice([0,0,1])
    args=["--backend",BACKEND,"--nbits",str(nb),"--pairs",str(pairs),"--steps",",".join(map(str,steps)),"--air",air,"--hashcommits",str(hc),"--shardsetup",str(ss),"--impols",str(im),"--boundaries",str(bd)]
    cmd=[sys.executable,"-m","torch.distributed.run","--nnodes=1","--nproc-per-node",str(world),"--master-addr","127.0.0.1","--master-port",str(port()),W,*args]
    r=subprocess.run(cmd,capture_output=True,text=True,timeout=600,env=dict(os.environ,OMP_NUM_THREADS="1"))
    ok = r.returncode==0 and r.stdout.count(" ok")==world
    n+=1; bad+= (not ok)
    print("%s world %d %s" % ("ok  " if ok else "FAIL", world, " ".join(args)), flush=True)
    if not ok: print(r.stdout[-1500:], r.stderr[-2500:], flush=True)
print("sharded fuzz: %d runs, %d failures, %.0f s" % (n,bad,time.time()-t0))
