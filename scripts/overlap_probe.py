"""How do the short, dependent set-up launches of one engine fare while another engine's pass streams the chip?

Engine A: the default benchmark shard (its pass kernel fills every CU for milliseconds).
Engine B: the same row count with few structural columns, so that a pass is short and the chained
set-up launches (k_chain_ratio / k_chain_finalize) dominate.  Each runs alone, then both at once from
two host threads on their own streams.  With an argument N the streams are kept apart: B's on the first N
CUs of the device, A's on the rest (LPAMD_STREAM_CU_RANGE).
"""
import sys, time, threading, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib
eng = importlib.import_module("linear-programming_amd.engine")

MAX_COST = 2
m, nA, nB = 32768, 65536, int(os.environ.get("NB", "4096"))
split = int(sys.argv[1]) if len(sys.argv) > 1 else 0
if split:
    os.environ["LPAMD_STREAM_CU_RANGE"] = "%d:256" % split
A = eng.TableauEngine(m, nA + m)
A.fill_synthetic(nA, 1)
if split:
    os.environ["LPAMD_STREAM_CU_RANGE"] = "0:%d" % split
B = eng.TableauEngine(m, nB + m)
B.fill_synthetic(nB, 2)
dA, dB = A.loop_depth(MAX_COST), B.loop_depth(MAX_COST)
print("CUs set aside for B:", split, "depths", dA, dB, flush=True)
A.run_pivots(MAX_COST, dA)
B.run_pivots(MAX_COST, dB)

def timed(e, k, out, key):
    t0 = time.perf_counter()
    res, done = e.run_pivots(MAX_COST, k)
    out[key] = (time.perf_counter() - t0, done, res)

kA, kB = dA * 20, dB * int(os.environ.get("PASSES_B", "60"))
r = {}
timed(A, kA, r, "A_alone")
timed(B, kB, r, "B_alone")
ta = threading.Thread(target=timed, args=(A, kA, r, "A_with_B"))
tb = threading.Thread(target=timed, args=(B, kB, r, "B_with_A"))
ta.start(); tb.start(); ta.join(); tb.join()
for k, (t, done, res) in r.items():
    print(f"{k}: {done} pivots in {t*1e3:.2f} ms = {done/t:.0f} pivots/s, {t/done*1e6:.1f} us/pivot (result {res})", flush=True)
