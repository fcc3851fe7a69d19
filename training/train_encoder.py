"""Entry-point shim with the reference's file name: ``torchrun training/train_encoder.py ...`` runs the MI355X
harness (omnibiote_amd/train_encoder.py) with the reference's flags (training/train_encoder.py:438-467)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnibiote_amd.train_encoder import parse_args, run  # noqa: E402

if __name__ == "__main__":
    run(parse_args())
