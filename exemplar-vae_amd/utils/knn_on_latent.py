"""kNN classification in the latent space (function names and arguments of reference utils/knn_on_latent.py:4-74).
The neighbour search is ONE launch of the fused distance + top-K kernel over the whole evaluation set; the label
vote is a vectorised count instead of a per-row loop."""
import torch

from evae import ops, shard

_NEIGHBOURS = 20          # neighbours kept per query; the k values voted on are prefixes of this list
_CLASSES = 10


def find_nearest_neighbors(z_val, z_train, z_train_log_var):
    """[len(z_val) x 20] training-row indices, nearest first.  `z_train_log_var` takes no part in the distance
    (reference :4-9 ignores it too)."""
    base = getattr(z_train, 'shard_base', None)
    if base is not None:      # z_train holds this rank's row block only: local top-20, one all-gather, exact merge
        return shard.sharded_topk(z_val, z_train, _NEIGHBOURS, index_base=int(base), sqrt=True)[0]
    return ops.pairdist_topk(z_val, z_train, _NEIGHBOURS, sqrt=True, want_val=False)[0]


class _RowBlock(torch.Tensor):
    """a plain tensor that remembers which global row its first row is (find_nearest_neighbors keeps its three arguments)"""
    shard_base = None


def extract_full_data(data_loader):
    """Whole content of a loader of (data, labels) or (data, indices, labels) batches -> (data, indices, labels);
    indices is an empty list when the loader carries none (reference :12-28)."""
    columns = {'data': [], 'indices': [], 'labels': []}
    for batch in data_loader:
        columns['data'].append(batch[0])
        columns['labels'].append(batch[-1])
        if len(batch) == 3:
            columns['indices'].append(batch[1])
    idx = columns['indices']
    return (torch.cat(columns['data'], dim=0), torch.cat(idx, dim=0) if idx else idx,
            torch.cat(columns['labels'], dim=0))


def _posterior_means(model, data, batch_size, sharded=False):
    """q(z|x) means of the full batches of `data` (a trailing partial batch is dropped, as in the reference).  sharded: only
    this rank's contiguous block of those rows is encoded; the result carries its first global row as .shard_base"""
    n = (len(data) // batch_size) * batch_size
    lo, hi = shard.shard_rows(n) if sharded else (0, n)
    chunks = [model.q_z(data[s:min(s + batch_size, hi)], prior=True)[0] for s in range(lo, hi, batch_size)]
    z = torch.cat(chunks, dim=0) if chunks else torch.zeros((0, model.args.z1_size), device=data.device)
    if sharded:
        z = z.as_subclass(_RowBlock)
        z.shard_base = lo
    return z


def report_knn_on_latent(train_loader, val_loader, test_loader, model, dir, knn_dictionary, args, val=True):
    """For every k in knn_dictionary: majority vote over the k nearest training latents, accuracy in percent
    (two decimals) appended to knn_dictionary[k].  val=True scores the validation split against the training split;
    val=False scores the test split against training + validation (reference :32-74)."""
    splits = [extract_full_data(loader) for loader in (train_loader, val_loader, test_loader)]
    (ref_x, _, ref_y), (val_x, _, val_y), (test_x, _, test_y) = splits
    ref_x, val_x = ref_x.to(args.device), val_x.to(args.device)
    if val is True:
        query_x, query_y = val_x, val_y
    else:
        ref_x, ref_y = torch.cat((ref_x, val_x), dim=0), torch.cat((ref_y, val_y), dim=0)
        query_x, query_y = test_x.to(args.device), test_y
    with torch.no_grad():
        sharded = bool(getattr(args, 'shard_exemplars', False)) and shard.is_active()
        z_ref = _posterior_means(model, ref_x, args.batch_size, sharded)
        z_query = _posterior_means(model, query_x, args.batch_size)
        neighbours = find_nearest_neighbors(z_query, z_ref, None).cpu()
    print(z_ref.shape)
    query_y = query_y[:len(neighbours)].long()
    classes = torch.arange(_CLASSES).view(1, 1, -1)
    for key in knn_dictionary.keys():
        k = int(key)
        votes = (ref_y[neighbours[:, :k]].reshape(len(neighbours), k, 1).long() == classes).sum(dim=1)
        hit = (votes.argmax(dim=1) == query_y).float().mean()
        acc = (hit * 10000).round().item() / 100
        print('K:', k, 'Accuracy:', acc)
        knn_dictionary[str(k)].append(acc)
