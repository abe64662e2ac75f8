"""Dotted-key prefix tree backing the class registry.

Same observable behaviour as the reference's ``utils/trie.py:4-32``: ``add`` splits the key on '.', creates
intermediate nodes and stores the value on the last one (re-adding overwrites); ``get`` returns the stored value,
or ``default_value`` when any path component is missing.  The reference also keeps a ``count`` that ``add``
increments and a successful ``get`` decrements (``trie.py:22,31``); that quirk is preserved because it is
observable.
"""


class Trie:
    class Node:
        __slots__ = ("value", "children")

        def __init__(self, value=None):
            self.value = value
            self.children = {}

    def __init__(self):
        self.root = Trie.Node()
        self.count = 0

    def _walk(self, key, create):
        node = self.root
        for part in key.split("."):
            nxt = node.children.get(part)
            if nxt is None:
                if not create:
                    return None
                nxt = node.children[part] = Trie.Node()
            node = nxt
        return node

    def add(self, key, value):
        self._walk(key, create=True).value = value
        self.count += 1

    def get(self, key, default_value=None):
        node = self._walk(key, create=False)
        if node is None:
            return default_value
        self.count -= 1
        return node.value
