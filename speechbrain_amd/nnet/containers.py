"""speechbrain.nnet.containers mirror: just enough container semantics for the ASR wiring
(nnet/containers.py:20-212, utils/callchains.py:17-19)."""
import inspect

import torch


class Sequential(torch.nn.ModuleDict):
    """Named sequential container; keyword layers keep their names as state_dict keys."""

    def __init__(self, *layers, input_shape=None, **named_layers):
        super().__init__()
        self.input_shape = input_shape
        self.length_layers = []
        for layer in layers:
            self.append(layer)
        for name, layer in named_layers.items():
            self.append(layer, layer_name=name)

    def append(self, layer, *args, layer_name=None, **kwargs):
        if layer_name is None:
            layer_name = str(len(self))
        elif layer_name in self:
            index = 0
            while f"{layer_name}_{index}" in self:
                index += 1
            layer_name = f"{layer_name}_{index}"
        if not isinstance(layer, torch.nn.Module):
            raise NotImplementedError("lazy (shape-inferred) layer construction is replaced by explicit modules here")
        try:
            takes_lengths = "lengths" in inspect.signature(layer.forward).parameters
        except (TypeError, ValueError):
            takes_lengths = False
        if takes_lengths:
            self.length_layers.append(layer_name)
        self.add_module(layer_name, layer)

    def forward(self, x):
        for layer in self.values():
            x = layer(x)
            if isinstance(x, tuple):
                x = x[0]
        return x


class LengthsCapableSequential(Sequential):
    """Passes ``lengths=`` to the children whose forward names that argument (containers.py:165-212)."""

    def forward(self, x, lengths=None):
        for name, layer in self.items():
            x = layer(x, lengths=lengths) if name in self.length_layers else layer(x)
            if isinstance(x, tuple):
                x = x[0]
        return x


class ModuleList(torch.nn.Module):
    """containers.py:215-260: plain list applied in order, children under ``layers``."""

    def __init__(self, *layers):
        super().__init__()
        self.layers = torch.nn.ModuleList(layers)

    def forward(self, x):
        for layer in self.layers:
            x = layer(x)
            if isinstance(x, tuple):
                x = x[0]
        return x

    def append(self, module):
        self.layers.append(module)
