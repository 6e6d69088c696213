"""The helpers the ETDiff CLI uses (reference helpers/utils.py:67-100, :182): seeding, argparse validators,
reverse_normalize, run ids.  Host-side glue."""
import argparse
import os
import random
import time

import numpy as np
import torch


def seed_everything(seed):
    random.seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def create_id():
    return int(time.time())


def exists(x):
    return x is not None


def is_positive_integer(value):
    iv = int(value)
    if iv <= 0:
        raise argparse.ArgumentTypeError(f"{value} is not a positive integer")
    return iv


def is_positive_float(value):
    fv = float(value)
    if fv <= 0:
        raise argparse.ArgumentTypeError(f"{value} is not a positive float")
    return fv


def is_file_on_disk(path):
    if not os.path.isfile(path):
        raise argparse.ArgumentTypeError(f"{path} is not a file on disk")
    return path


def is_file_not_on_disk(path):
    if os.path.isfile(path):
        raise argparse.ArgumentTypeError(f"{path} already exists")
    return path


def reverse_normalize(data, min, max):
    return data * (max - min) + min
