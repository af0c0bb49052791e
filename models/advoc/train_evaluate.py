#!/usr/bin/env python
"""Same entry point path as the reference (models/advoc/train_evaluate.py); the implementation
lives in advoc_amd/train_evaluate.py."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))

from advoc_amd.train_evaluate import main  # noqa: E402

if __name__ == '__main__':
  main()
