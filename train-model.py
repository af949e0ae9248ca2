#!/usr/bin/env python3
"""Drop-in for the reference's train-model.py on the MI355X path: same constants, same loop semantics
(music-style-transfer_amd/style/train.py), run from the repository root:

    python train-model.py [data_path]
"""
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, 'music-style-transfer_amd'))

from style.train import main  # noqa: E402

if __name__ == '__main__':
    main(*(sys.argv[1:2] or ['data/Lakh MIDI Dataset/clean_midi/']), n_iterations=5000, iter_size=2,
         training_info_path='training.csv', save_path='snapshots/', save_interval=100)
