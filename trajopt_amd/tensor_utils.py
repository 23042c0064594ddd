"""The one helper of the reference's `trajopt/tensor_utils.py` that the rollout path uses
(`stack_tensor_dict_list`, :71-86, called at `trajopt/utils.py:97`)."""
import numpy as np


def stack_tensor_list(tensor_list):
    return np.array(tensor_list)


def stack_tensor_dict_list(tensor_dict_list):
    """list of {key: array | nested dict} -> {key: stacked array | nested stacked dict}"""
    first = tensor_dict_list[0]
    out = {}
    for key, example in first.items():
        column = [d[key] for d in tensor_dict_list]
        out[key] = stack_tensor_dict_list(column) if isinstance(example, dict) else stack_tensor_list(column)
    return out
