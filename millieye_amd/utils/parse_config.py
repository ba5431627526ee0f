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
    for raw in text.split("\n"):
        line = raw.strip()
        if not line or line.startswith("#"):
            continue
        yield line


def parse_model_config(path):
    """cfg file -> list of dicts, first entry is the ``[net]`` hyper-parameter block."""
    with open(path, "r") as fh:
        text = fh.read()
    blocks = []
    for line in _significant_lines(text):
        if line.startswith("["):
            block = {"type": line[1:-1].rstrip()}
            if block["type"] == "convolutional":
                block["batch_normalize"] = 0  # int on purpose: falsy default, see reference :14-15
            blocks.append(block)
            continue
        key, value = line.split("=")  # exactly one '=' per line, like the reference
        blocks[-1][key.rstrip()] = value.strip()
    return blocks


def parse_data_config(path):
    """``key=value`` data file -> dict (space separated values become lists)."""
    options = {"gpus": "0,1,2,3", "num_workers": "10"}
    with open(path, "r") as fh:
        for raw in fh.readlines():
            line = raw.strip()
            if line == "" or line.startswith("#"):
                continue
            key, value = line.split("=")
            parts = value.split(" ")
            options[key.strip()] = parts if len(parts) > 1 else value
    return options
