import os, sys, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
from limitador_amd.engine import Engine
from limitador_amd.ingest import Ingest
from test_ingest_cpu import rls_request
rng=np.random.default_rng(5)
eng=Engine(capacity_cells=1<<22, max_batch_hits=1<<21, max_limits=64)
g=Ingest(); methods=["GET","POST","PUT"]
for n in range(4):
    for j in range(8):
        nv=0 if j<2 else (1 if j<5 else 2)
        conds=[f"descriptors[0]['method'] {'==' if j%2==0 else '!='} '{methods[j%3]}'"]
        variables=[] if nv==0 else (["descriptors[0]['user']"] if nv==1 else ["descriptors[0]['app']","descriptors[0]['user']"])
        g.add_limit(f"ns{n}", 10**9 if j==0 else 1000, [1,10,60,3600][(n+j)%4], conds, variables)
g.install(eng)
msgs=[rls_request(f"ns{int(rng.integers(0,4))}", [[("method",methods[int(rng.integers(0,3))]),("path","/x"),("user",f"user{int(rng.zipf(1.2))%200000}"),("app",f"app{int(rng.integers(0,5))}")]]) for _ in range(32768)]
prep=g.prepare_batch(msgs)
now=1_700_000_000_000_000
for hdr in (False, True):
    g.serve_prepared(eng, prep, now, with_headers=hdr)
    os.environ["RLI_TRACE"]="1"
    g.serve_prepared(eng, prep, now+1000, with_headers=hdr)
    del os.environ["RLI_TRACE"]
