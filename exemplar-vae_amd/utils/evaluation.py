"""Evaluation loops with the reference's signatures (reference utils/evaluation.py:11-140): ELBO against
the full exemplar cache and the IWAE test log-likelihood.  The prior over ALL training exemplars
([B x N_train], [S x N_train] with S = 5000 per test image) is where the fused prior kernel dominates;
sums and the final log-sum-exp stay on the device (one host read per loop instead of one per batch /
per image, reference :27-29,96)."""
import math
import time

import numpy as np
import torch

from evae import hostcpu, shard
from utils.utils import load_model


def load_all_pseudo_input(args, model, dataset):
    """Exemplar embedding of the whole training set: (z [N x z], logvar [N x z], arange(N)).  With args.shard_exemplars on a
    multi-rank run every rank encodes and keeps only its contiguous row block of the cache (models/BaseModel.py::cache_z_shard,
    SURVEY 8e): the loops below then score the same queries on every rank against 1/R of the exemplars each and merge the
    packed partial log-sum-exps with one all-gather per call."""
    if args.prior == 'exemplar_prior':
        if getattr(args, 'shard_exemplars', False) and shard.is_active():
            return model.cache_z_shard(dataset)
        exemplars_z, exemplars_log_var = model.cache_z(dataset)
        return (exemplars_z, exemplars_log_var, torch.arange(len(exemplars_z)))
    if args.prior == 'vampprior':
        pseudo_means = model.means(model.idle_input)
        if 'conv' in args.model_name:
            pseudo_means = pseudo_means.view(-1, args.input_size[0], args.input_size[1], args.input_size[2])
        return model.q_z(pseudo_means, prior=True)
    if args.prior == 'standard':
        return None
    raise Exception("wrong name of prior")


def evaluate_loss(args, model, loader, dataset=None, exemplars_embedding=None):
    hostcpu.limit_host_threads()
    model.eval()
    if exemplars_embedding is None:
        exemplars_embedding = load_all_pseudo_input(args, model, dataset)
    totals = torch.zeros(3, device=args.device, dtype=torch.float64)
    with shard.replicated_noise(model, exemplars_embedding, args.device):
        for batch in loader:
            data = batch[0].to(args.device)
            loss, RE, KL = model.calculate_loss((data, None), average=False, exemplars_embedding=exemplars_embedding)
            totals += torch.stack((loss.sum(), -RE.sum(), KL.sum())).double()
    elbo, re, kl = (totals / len(loader.dataset)).tolist()
    return elbo, re, kl


IWAE_ROWS_PER_LAUNCH = 20000     # importance samples scored per calculate_loss call (several test images at once)


def calculate_likelihood(args, model, loader, S=5000, exemplars_embedding=None):
    """IWAE estimate -mean_x [logsumexp_s(-loss_s) - log S]  (reference :72-103).  The reference scores one test image
    (S samples) per pass and takes the log-sum-exp on the host; here a pass carries as many images as fit in
    IWAE_ROWS_PER_LAUNCH rows -- the S copies of an image are consecutive rows, so the eps stream is consumed in the
    same order as image-by-image -- and the per-image log-sum-exp stays on the device."""
    hostcpu.limit_host_threads()
    dataset = loader.dataset
    group = max(1, min(IWAE_ROWS_PER_LAUNCH // max(int(S), 1), 64))
    aux = torch.utils.data.DataLoader(dataset, batch_size=group)
    n_img = len(dataset)
    out = torch.empty(n_img, device=args.device, dtype=torch.float64)
    t0 = time.time()
    done = 0
    with shard.replicated_noise(model, exemplars_embedding, args.device):
        for batch in aux:
            data = batch[0].to(args.device)
            data = data.reshape(data.size(0), -1)
            g = data.size(0)
            if done // 100 != (done + g) // 100 or done == 0:
                print(time.time() - t0)
                t0 = time.time()
                print('{:.2f}%'.format(done / (1. * n_img) * 100))
            prob = model.importance_sample_losses(data, S, exemplars_embedding)
            ll = torch.logsumexp(-prob.double().view(g, S), dim=1)
            if model.args.use_logit:
                lambd = model.args.lambd
                sp = torch.nn.functional.softplus
                ll = ll - (-sp(-data) - sp(data) - math.log((1 - 2 * lambd) / 256)).sum(dim=1).double()
            out[done:done + g] = ll - math.log(S)
            done += g
    return -float(out.mean().item())


def final_evaluation(train_loader, test_loader, valid_loader, best_model_path_load, model, optimizer, args, dir):
    _ = load_model(best_model_path_load, model, optimizer)
    model.eval()
    with torch.no_grad():
        emb = load_all_pseudo_input(args, model, train_loader.dataset)
        test_elbo, test_re, test_kl = evaluate_loss(args, model, test_loader, dataset=train_loader.dataset,
                                                    exemplars_embedding=emb)
        valid_elbo, _, _ = evaluate_loss(args, model, valid_loader, dataset=valid_loader.dataset,
                                         exemplars_embedding=emb)
        train_elbo, _, _ = evaluate_loss(args, model, train_loader, dataset=train_loader.dataset,
                                         exemplars_embedding=emb)
        test_log_likelihood = calculate_likelihood(args, model, test_loader, exemplars_embedding=emb, S=args.S)
    txt = ('FINAL EVALUATION ON TEST SET\nLogL (TEST): {:.2f}\nLogL (TRAIN): {:.2f}\nELBO (TEST): {:.2f}\n'
           'ELBO (TRAIN): {:.2f}\nELBO (VALID): {:.2f}\nRE: {:.2f}\nKL: {:.2f}').format(
        test_log_likelihood, 0, test_elbo, train_elbo, valid_elbo, test_re, test_kl)
    print(txt)
    with open(dir + 'vae_experiment_log.txt', 'a') as f:
        print(txt, file=f)
    torch.save(test_log_likelihood, dir + args.model_name + '.test_log_likelihood')
    torch.save(test_elbo, dir + args.model_name + '.test_loss')
    torch.save(test_re, dir + args.model_name + '.test_re')
    torch.save(test_kl, dir + args.model_name + '.test_kl')


def compute_mean_variance_per_dimension(args, model, test_loader):
    means = torch.cat([model.q_z(batch.to(args.device))[0] for batch, _ in test_loader], dim=0)
    active = int((means.var(dim=0, unbiased=False) > 0.01).sum().item())
    print('active dimensions', active)
    return active
