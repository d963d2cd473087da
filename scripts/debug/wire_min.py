import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from limitador_amd.engine import Engine
from limitador_amd.ingest import Ingest
from test_ingest_cpu import rls_request
for keys in ("exact", "hashed"):
    eng = Engine(capacity_cells=1 << 12, max_batch_hits=1 << 12)
    g = Ingest(keys=keys)
    g.add_limit("ns", 1, 60, ["descriptors[0]['method'] == 'PUT'"], [])
    g.add_limit("ns", 1, 60, ["descriptors[0]['method'] != 'PUT'"], [])
    g.add_limit("ns", 1, 60, [], ["descriptors[0]['user']"])
    g.install(eng)
    msgs = [rls_request("ns", [[("method", m), ("user", "u")]]) for m in ("PUT", "GET", "PUT", "GET")]
    print(keys, g.serve_batch(eng, msgs, 1_700_000_000_000_000)[0], [(hex(int(r["key"])), int(r["limit"]) & 0xFFFF, int(r["value"]), int(r["reserved"])) for r in eng.dump_cells()])
    g.close(); eng.close()
