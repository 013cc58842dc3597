"""The product's N>1 launch path, rehearsed on one GPU: the driver's own command line
`python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 ...` with
MMC_DIST_BACKEND=gloo, so that both ranks share cuda:0 and the final reduction runs over gloo (on
the 8-GPU node the same code runs one rank per GPU over RCCL).  Launched by tests/conftest.py
before this process touches the GPU; checked here:

  * the JSON contract line of rank 0 (n_gpus, moves, weak scaling, roofline object),
  * running totals equal a recompute on every rank (energy_drift_rel),
  * the 2 x 64 chains are THE SAME chains as one process running global indices 0..127:
    trajectories depend on (seed, global replica index) only -- identical accept counts and the
    same energy sum.

No scaling claim is made from this: two ranks share one device."""
import json
import os

import pytest

pytestmark = pytest.mark.gpu


def load(out, name):
    rc = open(os.path.join(out, name + ".rc")).read().strip()
    err = open(os.path.join(out, name + ".err")).read()
    assert rc == "0", (name, rc, err[-2000:])
    lines = [l for l in open(os.path.join(out, name + ".json")).read().splitlines() if l.startswith("{")]
    assert len(lines) == 1, (name, lines, err[-2000:])      # ONE JSON line, from rank 0 only
    return json.loads(lines[0])


def test_two_ranks_share_nothing_but_the_final_reduction(dist_rehearsal):
    two, one = load(dist_rehearsal, "two"), load(dist_rehearsal, "one")
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1
    assert two["scaling"] == "weak" and two["unit"] == "moves/s" and two["dtype"] == "f64"
    assert two["steps"] == 20 and two["warmup"] == 5
    assert two["config"]["replicas_per_gpu"] == 64 and two["config"]["replicas_total"] == 128
    assert one["config"]["replicas_total"] == 128
    moves = 2 * 64 * 20
    assert abs(two["value"] * two["ms_per_step"] * 1e-3 * 20 - moves) < 1e-6 * moves   # value = moves / time
    for d in (two, one):
        assert d["energy_drift_rel"] < 1e-12
        assert d["roofline"]["kernel"] in ("k_move_eval_wave", "k_move_eval_fast")
        assert d["roofline"]["frac"] > 0
        assert d["vs_baseline"] is None and d["higher_is_better"] is True
    # the same 128 chains either way
    assert two["acceptance"] == one["acceptance"]
    assert two["overlaps"] == one["overlaps"]
    e2, e1 = two["energy_mean_per_replica"], one["energy_mean_per_replica"]
    assert abs(e2 - e1) < 1e-13 * abs(e1), (e2, e1)
