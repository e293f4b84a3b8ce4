"""Model-name -> class mapping and atomic checkpoint helpers (reference utils/utils.py:4-32)."""
import os

import torch


def importing_model(args):
    name = args.model_name
    if name == 'vae':
        from models.VAE import VAE
    elif name == 'hvae_2level':
        from models.HVAE_2level import VAE
    elif name == 'convhvae_2level':
        from models.convHVAE_2level import VAE
    elif name == 'single_conv':
        from models.fully_conv import VAE
    else:
        # 'pixelcnn' / 'new_vae' are outside the accelerated path
        raise Exception('Wrong name of the model!')
    return VAE


def save_model(save_path, load_path, content):
    """Write to a temporary path, then rename: a crash never leaves a torn checkpoint."""
    torch.save(content, save_path)
    os.rename(save_path, load_path)


def load_model(load_path, model, optimizer=None):
    checkpoint = torch.load(load_path)
    model.load_state_dict(checkpoint['state_dict'])
    if optimizer is not None:
        optimizer.load_state_dict(checkpoint['optimizer'])
    return checkpoint
