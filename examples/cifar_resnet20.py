"""CIFAR-10 ResNet-20 (synthetic CIFAR-shaped data when the files are absent).

    python examples/cifar_resnet20.py -m c -cs async -n 8 -b 128 --dtype bf16     # async PS, GPU-resident shards
    python examples/cifar_resnet20.py -m d -ds keras -n 8 -b 128 --dtype bf16     # all-reduce comparison
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.cli.launcher import main  # noqa: E402
from distributed_tensorflow_amd.data.dataset import Dataset  # noqa: E402
from distributed_tensorflow_amd.data.sources import load_cifar10  # noqa: E402
from distributed_tensorflow_amd.models.resnet import resnet20  # noqa: E402


def model_fn():
    return resnet20(10)


def dataset_fn(batch_size, type='train', shard=True, index=0, buffer_size=10000, reshape=True, n_nodes=None):
    sizes = (int(os.environ.get("DTF_SYNTH_CIFAR_TRAIN", 50000)), int(os.environ.get("DTF_SYNTH_CIFAR_TEST", 10000)))
    (xtr, ytr), (xte, yte), _ = load_cifar10(synthetic_sizes=sizes)
    images, labels = (xtr, ytr) if type == 'train' else (xte, yte)
    images = (images.astype(np.float32) / np.float32(255) - np.float32(0.5)) / np.float32(0.25)   # NHWC
    ds = Dataset.from_tensor_slices((images, labels.astype(np.int32)))
    if shard:
        ds = ds.shard(n_nodes or int(os.environ.get("DTF_N_NODES", "1")), index)
    return ds.batch(batch_size).shuffle(buffer_size)


if __name__ == "__main__":
    sys.exit(main(model_fn, dataset_fn))
