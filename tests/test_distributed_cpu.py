"""N > 1 path of bench.py on CPU: world_size-2 gloo rendezvous on 127.0.0.1, disjoint window shards, max-over-ranks
timing and summed throughput (the data path itself has no collective: replicas only)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json
sys.path.insert(0, %r)
import __graft_entry__ as g
g.load_package()
from vins_mono_amd import dist_util as D, synth
rank, local, world = D.env_rank()
assert D.init("gloo")
seeds = D.window_seeds(rank, 4)
probs = [synth.SyntheticSequence(s, L=12).window(0) for s in seeds[:1]]
elapsed = 0.5 + 0.25 * rank                      # pretend rank 1 is slower
D.barrier()
tmax = D.max_over_ranks(elapsed)
total = D.sum_over_ranks(len(seeds))
D.barrier()
import sys
sys.stdout.write(json.dumps(dict(rank=rank, world=world, seeds=seeds, tmax=tmax, total=total, L=int(len(probs[0]["inv_depth"])))) + "\n")
sys.stdout.flush()
D.finish()
''' % ROOT


def test_two_rank_gloo_sharding_and_reductions(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29533", str(script)], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    import json
    rows, dec, txt, pos = [], json.JSONDecoder(), r.stdout, 0       # the two ranks share one pipe: tolerate glued lines
    while (pos := txt.find("{", pos)) >= 0:
        obj, pos = dec.raw_decode(txt, pos)
        rows.append(obj)
    assert len(rows) == 2
    rows.sort(key=lambda d: d["rank"])
    assert rows[0]["world"] == 2
    assert set(rows[0]["seeds"]).isdisjoint(rows[1]["seeds"]) and len(rows[0]["seeds"]) == 4
    assert all(abs(d["tmax"] - 0.75) < 1e-12 and d["total"] == 8 for d in rows)
