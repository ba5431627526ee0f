"""Darknet ``.cfg`` / ``.data`` text parsers.

Host-side mirror of the reference interface
``module3_our_dataset/utils/parse_config.py:3-21`` (``parse_model_config``) and
``:23-38`` (``parse_data_config``): same names, same return shapes, same quirks
(every value stays a *string*; a ``[convolutional]`` block gets the *integer*
default ``batch_normalize = 0``; ``.data`` files get the ``gpus`` /
``num_workers`` string defaults).
"""

__all__ = ["parse_model_config", "parse_data_config"]


def _significant_lines(text):
    """Stripped lines that are neither blank nor ``#`` comments."""
    return [ln for ln in (raw.strip() for raw in text.split("\n")) if ln and ln[0] != "#"]


def parse_model_config(path):
    """cfg file -> list of dicts, first entry is the ``[net]`` hyper-parameter block."""
    with open(path, "r") as fh:
        lines = _significant_lines(fh.read())
    blocks = []
    for line in lines:
        if line[0] == "[":
            kind = line[1:-1].rstrip()
            # ``batch_normalize`` gets an *int* 0 default (falsy) while parsed values stay strings, reference :14-15
            blocks.append({"type": kind, "batch_normalize": 0} if kind == "convolutional" else {"type": kind})
        else:
            key, value = line.split("=")  # exactly one '=' per line, like the reference
            blocks[-1][key.rstrip()] = value.strip()
    return blocks


def parse_data_config(path):
    """``key=value`` data file -> dict (space separated values become lists)."""
    options = dict(gpus="0,1,2,3", num_workers="10")
    with open(path, "r") as fh:
        entries = [raw.strip() for raw in fh.readlines()]
    for entry in entries:
        if entry == "" or entry.startswith("#"):
            continue
        key, value = entry.split("=")
        pieces = value.split(" ")
        options[key.strip()] = value if len(pieces) == 1 else pieces
    return options
