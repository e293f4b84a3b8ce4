"""One training epoch with the reference's signature (reference utils/training.py:5-51).

Differences that do not change results: losses are accumulated on the device and read back once per
epoch instead of three .item() syncs per step (training.py:42-44), and the exemplar images never leave
HBM (the model keeps a device-resident copy of dataset.tensors[0])."""
import torch


def set_beta(args, epoch):
    if args.warmup == 0:
        return 1.
    return min(1. * epoch / args.warmup, 1.)


def train_one_epoch(epoch, args, train_loader, model, optimizer):
    model.train()
    beta = set_beta(args, epoch)
    print('beta: {}'.format(beta))
    if args.approximate_prior is True:
        with torch.no_grad():
            cache = tuple(model.cache_z(train_loader.dataset))
    else:
        cache = None
    totals = None
    for data, indices, target in train_loader:
        data, indices = data.to(args.device), indices.to(args.device)
        x = torch.bernoulli(data) if args.dynamic_binarization else data
        optimizer.zero_grad()
        loss, RE, KL = model.calculate_loss((x, indices), beta, average=True, cache=cache,
                                            dataset=train_loader.dataset)
        loss.backward()
        optimizer.step()
        with torch.no_grad():
            step_vals = torch.stack((loss.detach(), -RE.detach(), KL.detach()))
            totals = step_vals if totals is None else totals + step_vals
            if cache is not None:
                cache = (cache[0].detach(), cache[1].detach())
    train_loss, train_re, train_kl = (totals / len(train_loader)).tolist()
    return train_loss, train_re, train_kl
