from .constants import loss_functions, torch_to_np_types, Loss
