"""Debug helper: replay tests/test_gpu_fuzz.py's sequence for one seed, comparing the table after EVERY step."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import oracle
from limitador_amd.engine import Engine
from limitador_amd.wire import RL_SIMPLE
import test_gpu_fuzz as F
import test_gpu_parity as P

seed = int(sys.argv[1])
made = []
def make_engine(capacity_cells=1 << 16, **kw):
    e = Engine(capacity_cells=capacity_cells, **kw); made.append(e); return e
orig_run_both = P.run_both
step_no = [0]
def checked(eng, orc, *a, **k):
    r = orig_run_both(eng, orc, *a, **k)
    return r
F.run_both = checked
# monkeypatch rng.choice of op to log and check state after each op: simplest is to copy the loop
rng = np.random.default_rng(1000 + seed)
src = open(F.__file__).read()
body = src[src.index("def test_random_operation_sequences"):]
body = body.replace("        now += int(rng.choice([0, 1, 1000, SEC // 2, 3 * SEC]))",
                    "        print('step', step, op, n, flush=True)\n        assert_same_state(eng, orc, n_simple_expected=None)\n        now += int(rng.choice([0, 1, 1000, SEC // 2, 3 * SEC]))")
ns = dict(F.__dict__)
exec(body.replace("@pytest.mark.parametrize(\"seed\", range(12))\n", ""), ns) if False else None
exec("import numpy as np\n" + body[body.index("def test_random"):] , ns)
ns["test_random_operation_sequences"](make_engine, seed)
print("ok")
