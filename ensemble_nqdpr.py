#!/usr/bin/env python3
"""Natural Questions ensemble -- same argv and output as the reference's MEVI/ensemble_nqdpr.py:254-270 (what
nqdpr_ensemble.sh runs): the marco ensemble with NQ's answer-based hit test (test_inverse_{offsets,array}.bin)."""
from argparse import ArgumentParser

from mevi_amd.metrics import ensemble_nqdpr_main

if __name__ == "__main__":
    parser = ArgumentParser()
    parser.add_argument("--dir_path", type=str, required=True)
    parser.add_argument("--ance_file", type=str, required=True)
    parser.add_argument("--fine_file", type=str, default=None)
    parser.add_argument("--coarse_file", type=str, default=None)
    parser.add_argument("--mapping_file", type=str, default=None)
    parser.add_argument("--alphas", type=str, default="0.4")
    parser.add_argument("--betas", type=str, default="0.03")
    parser.add_argument("--gammas", type=str, default="0.02")
    parser.add_argument("--recall_num", type=str, default="5,20,100")
    parser.add_argument("--ofile", type=str, default=None)
    parser.add_argument("--noensemble", action="store_true", default=False)
    a = parser.parse_args()
    ensemble_nqdpr_main(a.dir_path, a.ance_file, a.fine_file, a.coarse_file, a.mapping_file, a.alphas, a.betas, a.gammas,
                        a.recall_num, a.ofile, a.noensemble)
