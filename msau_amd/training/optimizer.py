"""`get_optimizer` counterpart (reference: model/training/optimizer.py:4-31): RMSprop by default, SGD+momentum or
Adam on request; `lr_decay_rate` is passed as weight decay, as the reference does."""
import torch


def get_optimizer(model, kwargs={}):
    params = model.parameters()
    name = kwargs.get("optimizer", "rmsprop")
    lr = kwargs.get("learning_rate", 0.001)
    wd = kwargs.get("lr_decay_rate", 0.0)
    if name == "momentum":
        opt = torch.optim.SGD(params, lr, kwargs.get("momentum", 0.9), weight_decay=wd)
    elif name == "rmsprop":
        opt = torch.optim.RMSprop(params, lr, weight_decay=wd)
    else:
        opt = torch.optim.Adam(params, lr, weight_decay=wd) if lr is not None else torch.optim.Adam(params)
    print("Optimizer: " + name)
    print("Learning Rate: " + ("" if lr is None else str(lr)))
    return opt
