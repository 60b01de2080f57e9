"""Launcher entry point with the example ``model_fn`` / ``dataset_fn``.

Usage mirrors the reference (reference README.md:10-21, initializer.py:70-184):

    python initializer.py -m c -cs sync  -n 3 -b 64 -ca y      # parameter server, synchronous
    python initializer.py -m c -cs async -n 4                  # parameter server, asynchronous
    python initializer.py -m d -ds keras  -n 8                 # mirrored all-reduce
    python initializer.py -m d -ds custom -n 8 --algorithm direct
    python initializer.py -m d -ds graph  -n 8 -d 4

Replace ``model_fn`` / ``dataset_fn`` below with your own to train something else.
"""
import os
import sys

import numpy as np

from distributed_tensorflow_amd.cli.launcher import main
from distributed_tensorflow_amd.data.dataset import Dataset
from distributed_tensorflow_amd.data.sources import load_mnist
from distributed_tensorflow_amd.models.mlp import reference_mlp


def model_fn():  # example function
    """Flatten(28, 28) -> Dense(512, relu) -> Dropout(0.2) -> Dense(10, softmax)."""
    return reference_mlp()


def dataset_fn(batch_size, type='train', shard=True, index=0, buffer_size=10000, reshape=True, n_nodes=None):
    """MNIST (or its synthetic stand-in when the files are absent): scaled to [0, 1] float32 with
    int32 labels, ``shard(n_nodes, index).batch(batch_size).shuffle(buffer_size)`` -- batches are
    shuffled, not examples, as in the reference."""
    sizes = (int(os.environ.get("DTF_SYNTH_MNIST_TRAIN", 60000)), int(os.environ.get("DTF_SYNTH_MNIST_TEST", 10000)))
    (xtr, ytr), (xte, yte), _ = load_mnist(synthetic_sizes=sizes)
    if type == 'train':
        images, labels = xtr, ytr
    elif type == 'test':
        images, labels = xte, yte
    else:
        raise ValueError('"' + str(type) + '" is an invalid dataset type')
    if reshape:
        images = images.reshape(images.shape[0], 28, 28, 1)
    images = images / np.float32(255)
    labels = labels.astype(np.int32)
    ds = Dataset.from_tensor_slices((images, labels))
    if shard:
        if n_nodes is None:
            n_nodes = int(os.environ.get("DTF_N_NODES", "1"))
        ds = ds.shard(n_nodes, index)
    return ds.batch(batch_size).shuffle(buffer_size)


if __name__ == '__main__':
    sys.exit(main(model_fn, dataset_fn))
