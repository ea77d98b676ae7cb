"""passport_configs/*.json -> nested passport_kwargs (reference experiments/utils.py:6-97).

A config is a tree whose leaves are `true`, `false` or an ASCII string (a signature to embed, which
also switches the layer on).  Every leaf becomes
    {'flag': bool, 'norm_type': ..., 'key_type': ..., 'sign_loss': sl_ratio [, 'b': string]}
and the dotted paths of the switched-on layers are returned when need_index=True (AlexNet: '4';
ResNet: 'layer4.0.convbnrelu_1').
"""


def _convert(node, path, norm_type, key_type, sl_ratio, keys):
    if isinstance(node, dict):
        return {k: _convert(v, path + [k], norm_type, key_type, sl_ratio, keys) for k, v in node.items()}
    signature = node if isinstance(node, str) else None
    flag = True if signature is not None else node
    if flag:
        keys.append('.'.join(path))
    leaf = {'flag': flag, 'norm_type': norm_type, 'key_type': key_type, 'sign_loss': sl_ratio}
    if signature is not None:
        leaf['b'] = signature
    return leaf


def _build(config, norm_type, key_type, sl_ratio, need_index):
    keys = []
    kwargs = {k: _convert(v, [k], norm_type, key_type, sl_ratio, keys) for k, v in config.items()}
    return (kwargs, keys) if need_index else kwargs


def construct_passport_kwargs(self, need_index=False):
    """`self` is an experiment object with passport_config / norm_type / key_type / sl_ratio attributes."""
    return _build(self.passport_config, self.norm_type, self.key_type, self.sl_ratio, need_index)


def construct_passport_kwargs_from_dict(self, need_index=False):
    """Same, from a plain dict with those four keys."""
    return _build(self['passport_config'], self['norm_type'], self['key_type'], self['sl_ratio'], need_index)
