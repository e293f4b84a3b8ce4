"""Latent-space kNN with the reference's signatures (reference utils/knn_on_latent.py:4-74) on the
fused distance + top-K kernel."""
import torch

from evae import ops


def find_nearest_neighbors(z_val, z_train, z_train_log_var):
    """Indices [len(z_val) x 20] of the 20 nearest training latents, nearest first; the third argument is
    ignored, as in the reference (:4-9)."""
    idx, _ = ops.pairdist_topk(z_val, z_train, 20, sqrt=True, want_val=False)
    return idx


def extract_full_data(data_loader):
    """Concatenate a loader of (data, [indices,] labels) batches (:12-28)."""
    datas, labels, indices = [], [], []
    for batch in data_loader:
        if len(batch) == 3:
            d, i, l = batch
            indices.append(i)
        else:
            d, l = batch
        datas.append(d)
        labels.append(l)
    full_indices = torch.cat(indices, dim=0) if len(indices) > 0 else indices
    return torch.cat(datas, dim=0), full_indices, torch.cat(labels, dim=0)


def report_knn_on_latent(train_loader, val_loader, test_loader, model, dir, knn_dictionary, args, val=True):
    """kNN label vote on q(z|x) means for k in knn_dictionary (:32-74); appends the accuracy (%)."""
    train_data, _, train_labels = extract_full_data(train_loader)
    val_data, _, val_labels = extract_full_data(val_loader)
    test_data, _, test_labels = extract_full_data(test_loader)
    train_data = train_data.to(args.device)
    val_data = val_data.to(args.device)
    if val is True:
        data_to_evaluate, labels = val_data, val_labels
    else:
        train_data = torch.cat((train_data, val_data), dim=0)
        train_labels = torch.cat((train_labels, val_labels), dim=0)
        data_to_evaluate, labels = test_data.to(args.device), test_labels
    bs = args.batch_size
    with torch.no_grad():
        z_train = torch.cat([model.q_z(train_data[i * bs:(i + 1) * bs], prior=True)[0]
                             for i in range(len(train_data) // bs)], dim=0)
        n_eval = (len(data_to_evaluate) // bs) * bs
        z_eval = torch.cat([model.q_z(data_to_evaluate[i * bs:(i + 1) * bs], prior=True)[0]
                            for i in range(len(data_to_evaluate) // bs)], dim=0)
        indices = find_nearest_neighbors(z_eval, z_train, None).cpu()   # one scan for the whole eval set
    print(z_train.shape)
    labels = labels[:n_eval]
    for k in knn_dictionary.keys():
        k = int(k)
        k_labels = train_labels[indices[:, :k]].reshape(len(indices), k).long()
        num_classes = 10
        counts = torch.stack([(k_labels == c).sum(dim=1) for c in range(num_classes)], dim=1)
        y_pred = torch.argmax(counts, dim=1)
        acc = (torch.mean((y_pred == labels.long()).float()) * 10000).round().item() / 100
        print('K:', k, 'Accuracy:', acc)
        knn_dictionary[str(k)].append(acc)
