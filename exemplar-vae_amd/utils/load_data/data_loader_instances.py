"""The datasets of the reference (utils/load_data/data_loader_instances.py:8-184) behind the same class names and the
same `load_dataset(args, ...)` entry point -- read from the files a download leaves on disk, with numpy / scipy only
(no torchvision, no network): datasets/<dataset_name>/... exactly where the reference's torchvision calls put them."""
import gzip
import os
import pickle
import struct
from types import SimpleNamespace

import numpy as np
import torch

from .base_load_data import base_load_data


def _root(args):
    return os.path.join('datasets', args.dataset_name)


def _need(path, hint):
    if not os.path.exists(path):
        raise FileNotFoundError("%s not found (%s); this build does not download datasets" % (path, hint))
    return path


def _first_existing(*paths):
    for p in paths:
        if os.path.exists(p):
            return p
    return paths[0]


def _read_idx(path):
    """IDX file (optionally .gz) -> uint8 ndarray of its declared shape"""
    opener = gzip.open if path.endswith('.gz') else open
    with opener(_need(path, "IDX file of the MNIST family"), 'rb') as f:
        zero, dtype_code, ndim = struct.unpack('>HBB', f.read(4))
        assert zero == 0 and dtype_code == 0x08, "unsupported IDX header in %s" % path
        shape = struct.unpack('>' + 'I' * ndim, f.read(4 * ndim))
        return np.frombuffer(f.read(), dtype=np.uint8).reshape(shape)


def _mnist_family(folder, raw_subdir):
    """(train, test) objects with the .data / .train_labels / .test_labels the base class expects"""
    def find(stem):
        raw = os.path.join(folder, raw_subdir, 'raw')
        return _first_existing(os.path.join(raw, stem), os.path.join(raw, stem + '.gz'), os.path.join(folder, stem))
    tensor = lambda stem: torch.from_numpy(_read_idx(find(stem)).copy())
    train = SimpleNamespace(data=tensor('train-images-idx3-ubyte'), train_labels=tensor('train-labels-idx1-ubyte'))
    test = SimpleNamespace(data=tensor('t10k-images-idx3-ubyte'), test_labels=tensor('t10k-labels-idx1-ubyte'))
    return train, test


class dynamic_mnist_loader(base_load_data):
    def obtain_data(self):
        return _mnist_family(_root(self.args), 'MNIST')


class fashion_mnist_loader(base_load_data):
    def obtain_data(self):
        return _mnist_family(_root(self.args), 'FashionMNIST')


class svhn_loader(base_load_data):
    def obtain_data(self):
        from scipy.io import loadmat

        def split(name):
            m = loadmat(_need(os.path.join(_root(self.args), name + '_32x32.mat'), "SVHN cropped-digits file"))
            labels = m['y'].astype(np.int64).squeeze()
            labels[labels == 10] = 0                                  # the file stores digit 0 as class 10
            return SimpleNamespace(data=np.transpose(m['X'], (3, 2, 0, 1)), labels=labels)     # -> [N x 3 x 32 x 32]
        return split('train'), split('test')

    def seperate_data_from_label(self, train_dataset, test_dataset):
        return (train_dataset.data, train_dataset.labels.astype(dtype=int),
                test_dataset.data, test_dataset.labels.astype(dtype=int))


class static_mnist_loader(base_load_data):
    """Larochelle's fixed binarisation: three .amat text files, no labels"""

    def obtain_data(self):
        def amat(split):
            path = _need(os.path.join(_root(self.args), 'binarized_mnist_%s.amat' % split), "binarized MNIST text file")
            return np.loadtxt(path, dtype=np.float32)
        x_train, x_val, x_test = amat('train'), amat('valid'), amat('test')
        no_labels = lambda x: np.zeros((x.shape[0], 1)).astype(int)
        return (x_train, x_val, no_labels(x_train), no_labels(x_val)), (x_test, no_labels(x_test))

    def seperate_data_from_label(self, train_dataset, test_dataset):
        x_train, x_val, y_train, y_val = train_dataset
        x_test, y_test = test_dataset
        return (x_train, x_val), (y_train, y_val), x_test, y_test

    def preprocessing_(self, x_train, x_test):
        return x_train, x_test


class omniglot_loader(base_load_data):
    def obtain_data(self):
        from scipy.io import loadmat
        raw = loadmat(_need(os.path.join(_root(self.args), 'chardata.mat'), "OMNIGLOT chardata.mat of the IWAE repository"))
        # stored column-major per character: 28 x 28 Fortran order -> row-major pixels
        pixels = lambda key: raw[key].T.astype('float32').reshape((-1, 28, 28)).reshape((-1, 28 * 28), order='F')
        return ((pixels('data'), raw['targetchar'].reshape((-1, 1))),
                (pixels('testdata'), raw['testtargetchar'].reshape((-1, 1))))

    def seperate_data_from_label(self, train_dataset, test_dataset):
        return (*train_dataset, *test_dataset)

    def preprocessing_(self, x_train, x_test):
        return x_train, x_test


class cifar10_loader(base_load_data):
    def obtain_data(self):
        folder = os.path.join(_root(self.args), 'cifar-10-batches-py')

        def batches(names):
            rows = []
            for n in names:
                with open(_need(os.path.join(folder, n), "CIFAR-10 python batch"), 'rb') as f:
                    rows.append(pickle.load(f, encoding='latin1')['data'])
            return SimpleNamespace(data=np.concatenate(rows).reshape(-1, 3, 32, 32).transpose(0, 2, 3, 1))   # NHWC like torchvision
        return batches(['data_batch_%d' % i for i in range(1, 6)]), batches(['test_batch'])

    def seperate_data_from_label(self, train_dataset, test_dataset):
        to_nchw = lambda d: np.swapaxes(np.swapaxes(d, 1, 2), 1, 3)
        no_labels = lambda d: np.zeros((d.shape[0], 1)).astype(int)
        tr, te = to_nchw(train_dataset.data), to_nchw(test_dataset.data)
        return tr, no_labels(tr), te, no_labels(te)


# dataset_name -> (loader class, default training_set_size or None, input_size, how the input type is decided)
_DATASETS = {
    'static_mnist': (static_mnist_loader, None, [1, 28, 28], 'binary'),
    'dynamic_mnist': (dynamic_mnist_loader, 50000, [1, 28, 28], 'grey_or_dynamic'),
    'fashion_mnist': (fashion_mnist_loader, 50000, [1, 28, 28], 'grey_or_dynamic'),
    'omniglot': (omniglot_loader, 23000, [1, 28, 28], 'dynamic'),
    'svhn': (svhn_loader, 60000, [3, 32, 32], 'continuous'),
    'cifar10': (cifar10_loader, 40000, [3, 32, 32], 'continuous'),
}


def load_dataset(args, training_num=None, use_fixed_validation=False, no_binarization=False, **kwargs):
    if training_num is not None:
        args.training_set_size = training_num
    if args.dataset_name not in _DATASETS:
        raise Exception('Wrong name of the dataset!')
    cls, default_n, input_size, kind = _DATASETS[args.dataset_name]
    args.input_size = list(input_size)
    fixed_size = kind == 'continuous'                 # svhn / cifar10 ignore training_num, as the reference does
    if default_n is not None and (training_num is None or fixed_size):
        args.training_set_size = default_n
    ctor = {}
    if kind == 'binary':
        args.input_type = 'binary'
    elif kind == 'continuous':
        args.input_type = 'continuous'
    elif kind == 'dynamic':
        args.input_type, args.dynamic_binarization = 'binary', True
    else:                                              # the two MNISTs: grey levels on request, else re-binarised per step
        if args.continuous is True:
            if args.dataset_name == 'fashion_mnist':
                print("*****Continuous Data*****")
            args.input_type, args.dynamic_binarization, no_binarization = 'gray', False, True
        else:
            args.input_type, args.dynamic_binarization = 'binary', True
        ctor = dict(use_fixed_validation=use_fixed_validation, no_binarization=no_binarization)
    train_loader, val_loader, test_loader, args = cls(args, **ctor).load_dataset(**kwargs)
    print('train size', len(train_loader.dataset))
    if val_loader is not None:
        print('val size', len(val_loader.dataset))
    print('test size', len(test_loader.dataset))
    return train_loader, val_loader, test_loader, args
