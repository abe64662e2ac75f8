"""JSON / pickle helpers (reference ``utils/file_utils.py``)."""
import json
import pickle
from pathlib import Path


def ensure_dir(dirname):
    Path(dirname).mkdir(parents=True, exist_ok=True)


def load_json(file_name):
    with open(file_name, "r") as f:
        return json.load(f)


def save_json(obj, file_name):
    with open(file_name, "w") as f:
        json.dump(obj, f, indent=4, sort_keys=False)


def load_pkl(file_name):
    with open(file_name, "rb") as f:
        return pickle.load(f)


def save_pkl(obj, file_name):
    with open(file_name, "wb") as f:
        pickle.dump(obj, f, pickle.HIGHEST_PROTOCOL)
