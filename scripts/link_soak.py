#!/usr/bin/env python3
"""Soak of the exchange inside the launches between PROCESSES that share GPU 0 (tests/link_two_process_worker.py, gloo for
the hosted collective): thousands of MAX_COST pivots (and MAX_REDUCTION pivots behind them) over 2..8 ranks, chains of 2..26 steps, HBM-sized and cache-sized shards;
every rank's pivots and right-hand side against the single-engine loop on the same GPU.

  python scripts/link_soak.py
"""
import importlib, json, os, socket, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lp = importlib.import_module("linear-programming_amd")
def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p
def run(world, m, n, seed, legs, env_extra):
    env = dict(os.environ, LPAMD_SHARD_LINK="1", **env_extra)
    port = free_port()
    d = tempfile.mkdtemp()
    outs = [os.path.join(d, "r%d.json" % r) for r in range(world)]
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "link_two_process_worker.py"), str(r), str(world), str(port), outs[r],
                               str(m), str(n), str(seed), ",".join(str(k) for k in legs)], env=env) for r in range(world)]
    codes = [p.wait(timeout=500) for p in procs]
    seen = [json.load(open(f)) for f in outs]
    return codes, seen
want = {}
def reference(m, n, seed, total, tail=0):
    key = (m, n, seed, total, tail)
    if key not in want:
        eng = lp.TableauEngine(m, n + m); eng.fill_synthetic(n, seed)
        res = eng.run_pivots(lp.MAX_COST, total)
        if tail:
            res = eng.run_pivots(lp.MAX_REDUCTION, tail)
        want[key] = ([list(t) for t in eng.trace()], [float(v).hex() for v in eng.read_column(n + m)], res)
        eng.close()
    return want[key]
cases = [(4, 32768, 65536, 1, (100,), {"LPAMD_TEST_MR_TAIL": "700"}), (3, 4096, 8192, 5, (10,), {"LPAMD_TEST_MR_TAIL": "1500"}),
         (4, 32768, 65536, 1, (1500, 1500), {}), (2, 32768, 65536, 1, (777, 778), {"LPAMD_SHARD_CHAIN": "32"}), (3, 32768, 65536, 2, (1000,), {"LPAMD_SHARD_CHAIN": "5"}),
         (2, 4096, 8192, 3, (3000,), {"LPAMD_SHARD_CHAIN": "2"}), (8, 4096, 8192, 3, (3000,), {}), (2, 4096, 8192, 3, (2500, 1, 1, 7), {"LPAMD_SHARD_CHAIN": "3"})]
for world, m, n, seed, legs, env in cases:
    codes, seen = run(world, m, n, seed, legs, env)
    tr, rhs, res = reference(m, n, seed, sum(legs), int(env.get("LPAMD_TEST_MR_TAIL", "0")))
    ok = codes == [0] * world and all("error" not in s and s["link"] and s["trace"] == tr and s["rhs"] == rhs for s in seen)
    print("world", world, "m", m, "n", n, "legs", legs, env, "chain", seen[0].get("chain"), "->", "same pivots and right-hand side on every rank" if ok else "DIFFERENT", [s.get("error") for s in seen if "error" in s], "single engine:", res, flush=True)
