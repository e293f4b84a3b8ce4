"""From raw dataset arrays to the three loaders the training / evaluation loops consume.

Behavioural contract: reference utils/load_data/base_load_data.py:8-120 (class and method names, the order of the
random draws, the dtypes and shapes handed to the model):
  * training set  -> TensorDataset(x float32 [N x D], index int64 [N x 1], label), DataLoader(shuffle=True)
  * validation    -> TensorDataset(x, label), DataLoader(shuffle=True);  test -> the same with shuffle=False
  * x in [0, 1]: /255 for 8-bit grey inputs, (x + 0.5)/256 for `continuous`, logit(lambd + (1 - 2 lambd)(x + u)/256)
    for `use_logit`; validation / test of dynamically binarised datasets are binarised ONCE with numpy seed 777
  * the training rows are what the model keeps resident in HBM (models.BaseModel.resident_data_ext): exemplars are
    gathered from it by index, the captured training step gathers its batches from it too (evae/graph.py).
Nothing here downloads anything: subclasses read the files a previous download left under datasets/<name>/."""
from abc import ABC, abstractmethod

import numpy as np
import torch
import torch.utils.data as data_utils

_TEST_BINARISATION_SEED = 777


class base_load_data(ABC):
    def __init__(self, args, use_fixed_validation=False, no_binarization=False):
        self.args = args
        self.train_num = args.training_set_size
        self.use_fixed_validation = use_fixed_validation
        self.no_binarization = no_binarization

    # ---- what a concrete dataset provides ------------------------------------------------------------------------
    @abstractmethod
    def obtain_data(self):
        """-> (train, test) in whatever form seperate_data_from_label of the same class understands"""

    def seperate_data_from_label(self, train_dataset, test_dataset):
        """default: objects with .data / .train_labels / .test_labels tensors (the MNIST family)"""
        as_int = lambda t: t.numpy().astype(int)
        return (train_dataset.data.numpy(), as_int(train_dataset.train_labels),
                test_dataset.data.numpy(), as_int(test_dataset.test_labels))

    # ---- value range -----------------------------------------------------------------------------------------------
    def logit(self, x):
        return np.log(x) - np.log1p(-x)

    def _to_unit_range(self, x):
        a = self.args
        if a.input_type not in ('gray', 'continuous'):
            return x / 255.
        if a.use_logit:                      # dequantise with uniform noise, squeeze into (lambd, 1 - lambd), logit
            return self.logit(a.lambd + (1 - 2 * a.lambd) * (x + np.random.rand(*x.shape)) / 256.)
        if a.continuous:
            return np.clip((x + 0.5) / 256., 0., 1.)
        return x

    def preprocessing_(self, x_train, x_test):
        return self._to_unit_range(x_train), self._to_unit_range(x_test)       # train first: the noise draws keep their order

    def binarize(self, x_val, x_test):
        """one fixed Bernoulli draw of the evaluation splits (the training split is re-drawn every step)"""
        self.args.input_type = 'binary'
        np.random.seed(_TEST_BINARISATION_SEED)
        return np.random.binomial(1, x_val), np.random.binomial(1, x_test)

    # ---- vampprior bookkeeping (outside the accelerated path, kept for API parity) ---------------------------------
    def vampprior_initialization(self, x_train, init_mean, init_std):
        a = self.args
        if a.use_training_data_init == 1:
            a.pseudoinputs_std = 0.01
            seed_rows = x_train[0:a.number_components].T
            noise = a.pseudoinputs_std * np.random.randn(np.prod(a.input_size), a.number_components)
            a.pseudoinputs_mean = torch.from_numpy(seed_rows + noise).float()
        else:
            a.pseudoinputs_mean, a.pseudoinputs_std = init_mean, init_std

    # ---- loaders -----------------------------------------------------------------------------------------------------
    def post_processing(self, x_train, x_val, x_test, y_train, y_val, y_test, init_mean=0.05, init_std=0.01, **kwargs):
        a = self.args
        as_x = lambda arr: torch.from_numpy(arr).float()
        row_ids = torch.from_numpy(np.arange(len(x_train)).reshape(-1, 1))
        train = data_utils.TensorDataset(as_x(x_train), row_ids, torch.from_numpy(y_train))
        train_loader = data_utils.DataLoader(train, batch_size=a.batch_size, shuffle=True, **kwargs)

        def eval_loader(x, y, shuffle):
            ds = data_utils.TensorDataset(as_x(x), torch.from_numpy(y))
            return data_utils.DataLoader(ds, batch_size=a.test_batch_size, shuffle=shuffle, **kwargs)
        val_loader = eval_loader(x_val, y_val, True) if len(x_val) > 0 else None
        test_loader = eval_loader(x_test, y_test, False)
        self.vampprior_initialization(x_train, init_mean, init_std)
        return train_loader, val_loader, test_loader

    def load_dataset(self, **kwargs):
        a = self.args
        x_train, y_train, x_test, y_test = self.seperate_data_from_label(*self.obtain_data())
        x_train, x_test = self.preprocessing_(x_train, x_test)
        if self.use_fixed_validation is False:                     # one global-generator shuffle before the split
            order = np.arange(len(x_train))
            np.random.shuffle(order)
            x_train, y_train = x_train[order], y_train[order]
        if a.dataset_name == 'static_mnist':                       # ships with its own validation split
            (x_train, x_val), (y_train, y_val) = x_train, y_train
        else:
            cut = self.train_num
            x_train, x_val = x_train[:cut], x_train[cut:]
            y_train, y_val = y_train[:cut], y_train[cut:]
        flat = int(np.prod(a.input_size))
        x_train, x_val, x_test = (np.reshape(x, (-1, flat)) for x in (x_train, x_val, x_test))
        if a.dynamic_binarization and self.no_binarization is False:
            x_val, x_test = self.binarize(x_val, x_test)
        print("data stats:")
        for x, y in ((x_train, y_train), (x_val, y_val), (x_test, y_test)):
            print(len(x), len(y))
        loaders = self.post_processing(x_train, x_val, x_test, y_train, y_val, y_test, **kwargs)
        return (*loaders, a)
