#!/usr/bin/env python3
"""BASELINE.json configs[4] ("cfg5"): writes the precision-study JSON from tests/cfg5_study.py --
the code tests/test_gpu_at_size.py::test_cfg5_tip3p_5000_precision_study_against_the_oracle runs
with fewer moves.  Needs a GPU and the oracle (a measurement script, not product code).

    python3 scripts/precision_study.py [--moves 10000] [--out gpurun_out/round4_cfg5_precision_study.json]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import cfg5_study
    ap = argparse.ArgumentParser()
    ap.add_argument("--moves", type=int, default=10000)
    ap.add_argument("--n-mol", type=int, default=cfg5_study.N_MOL)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "round4_cfg5_precision_study.json"))
    args = ap.parse_args()
    out = cfg5_study.run(args.moves, args.n_mol)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
