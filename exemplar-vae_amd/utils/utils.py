"""Model registry and checkpoint helpers (call signatures of reference utils/utils.py:4-32)."""
import importlib
import os

import torch

# model_name -> module that defines the class `VAE`.  'pixelcnn' / 'new_vae' of the reference are outside the
# accelerated path.
_MODEL_MODULES = {
    'vae': 'models.VAE',
    'hvae_2level': 'models.HVAE_2level',
    'convhvae_2level': 'models.convHVAE_2level',
    'single_conv': 'models.fully_conv',
}


def importing_model(args):
    module = _MODEL_MODULES.get(args.model_name)
    if module is None:
        raise Exception('Wrong name of the model!')
    return importlib.import_module(module).VAE


def save_model(save_path, load_path, content):
    """Checkpoint written under `save_path`, then moved onto `load_path` in one step: readers never see a torn file."""
    torch.save(content, save_path)
    os.replace(save_path, load_path)


def load_model(load_path, model, optimizer=None):
    state = torch.load(load_path)
    model.load_state_dict(state['state_dict'])
    if optimizer is not None:
        optimizer.load_state_dict(state['optimizer'])
    return state
