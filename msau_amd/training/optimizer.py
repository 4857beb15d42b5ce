"""`get_optimizer` counterpart (reference: model/training/optimizer.py:4-31).

Table-driven: the option names and their defaults are the reference's -- `optimizer` in {"momentum", "rmsprop"
(default), anything else = Adam}, `learning_rate` 1e-3, `momentum` 0.9 -- and so is its quirk of handing
`lr_decay_rate` to the optimizer as *weight decay*.  Prints the same two lines the reference prints."""
import torch

_DEFAULTS = {"optimizer": "rmsprop", "learning_rate": 1e-3, "lr_decay_rate": 0.0, "momentum": 0.9}


def _adam(parameters, opt):
    if opt["learning_rate"] is None:                 # the reference then falls back to torch's own defaults
        return torch.optim.Adam(parameters)
    return torch.optim.Adam(parameters, lr=opt["learning_rate"], weight_decay=opt["lr_decay_rate"])


_FACTORIES = {
    "momentum": lambda parameters, opt: torch.optim.SGD(parameters, lr=opt["learning_rate"], momentum=opt["momentum"],
                                                        weight_decay=opt["lr_decay_rate"]),
    "rmsprop": lambda parameters, opt: torch.optim.RMSprop(parameters, lr=opt["learning_rate"],
                                                           weight_decay=opt["lr_decay_rate"]),
}


def get_optimizer(model, kwargs={}):
    opt = {key: kwargs.get(key, default) for key, default in _DEFAULTS.items()}
    made = _FACTORIES.get(opt["optimizer"], _adam)(model.parameters(), opt)
    shown = "" if opt["learning_rate"] is None else str(opt["learning_rate"])
    print(f"Optimizer: {opt['optimizer']}")
    print(f"Learning Rate: {shown}")
    return made
