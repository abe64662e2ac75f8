"""String-keyed plugin registry: ``register_cls("model.ResNet")`` / ``find_cls("model.ResNet")``.

Mirrors the reference's ``utils/class_registry.py:4-14``; the keys used on the inference path are
``model.ResNet``, ``model.CNN``, ``data_loader.AudioDataLoader``, ``loss_fn.*``, ``metric.Acc``,
``metric.PerClassAcc`` and ``dataset.*`` (resolved only through ``find_cls(f"{kind}.{name}")``,
reference ``run/test.py:61,82,86`` and ``run/run_utils.py:38,46``).
"""
from .trie import Trie

_REGISTRY = Trie()


def register_cls(identifier):
    def decorator(obj):
        _REGISTRY.add(identifier, obj)
        return obj
    return decorator


def find_cls(identifier, default_value=None):
    return _REGISTRY.get(identifier, default_value)


def install_into(register_fn):
    """Register every honk2_amd plugin into ANOTHER registry (e.g. the reference's own ``utils.register_cls``)
    so that unmodified honk2 configs resolve to the MI355X-native classes.  See INTEGRATION.md."""
    def walk(node, prefix):
        for name, child in node.children.items():
            key = f"{prefix}.{name}" if prefix else name
            if child.value is not None:
                register_fn(key)(child.value)
            walk(child, key)
    walk(_REGISTRY.root, "")
