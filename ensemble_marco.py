#!/usr/bin/env python3
"""Ensemble of the dense list, the NCI fine list and the beam-cluster ranks -- same argv and
output as the reference's MEVI/ensemble_marco.py:243-259 (what marco_ensemble.sh runs)."""
from argparse import ArgumentParser

from mevi_amd.metrics import ensemble_main

if __name__ == "__main__":
    parser = ArgumentParser()
    parser.add_argument("--dir_path", type=str, default=None)
    parser.add_argument("--gt_file", type=str, required=True)
    parser.add_argument("--ance_file", type=str, required=True)
    parser.add_argument("--fine_file", type=str, default=None)
    parser.add_argument("--coarse_file", type=str, default=None)
    parser.add_argument("--mapping_file", type=str, default=None)
    parser.add_argument("--alphas", type=str, default="0.6")
    parser.add_argument("--betas", type=str, default="0.03")
    parser.add_argument("--gammas", type=str, default="0.02")
    parser.add_argument("--recall_num", type=str, default="10,50,1000")
    parser.add_argument("--ofile", type=str, default=None)
    a = parser.parse_args()
    ensemble_main(a.dir_path, a.gt_file, a.ance_file, a.fine_file, a.coarse_file, a.mapping_file,
                  a.alphas, a.betas, a.gammas, a.recall_num, a.ofile)
    from mevi_amd.phases import finish

    finish()
