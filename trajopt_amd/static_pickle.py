"""Static reader for the data held in pickle files.

The reference ships its only known-answer data (stored MPPI agents with real
MuJoCo transitions) as protocol-3 pickles.  Running the pickle virtual machine
on an untrusted file imports modules and calls objects the file names, so this
module never does: it walks the opcode stream with ``pickletools.genops`` (a
pure disassembler), keeps a symbolic stack of *inert* nodes and only ever
materialises bytes, ints, floats, strings, lists, tuples and dicts.  Class
references (``GLOBAL``), constructor calls (``NEWOBJ``/``REDUCE``) and state
assignment (``BUILD``) become plain records; nothing named by the file is
imported, looked up or called.

``decode`` then turns the numpy idioms of the pickle protocols 2 to 5 into arrays
with ``numpy.frombuffer`` on the raw byte payload, under a whitelist of dtype
strings.
"""
import pickletools

import numpy as np


class Global:
    __slots__ = ("name",)

    def __init__(self, name):
        self.name = name

    def __repr__(self):
        return "Global(%s)" % self.name


class Inert:
    """Record of a NEWOBJ / REDUCE: what would have been called, never called."""
    __slots__ = ("callee", "args", "state")

    def __init__(self, callee, args):
        self.callee, self.args, self.state = callee, args, None

    def __repr__(self):
        return "Inert(%r)" % (self.callee,)


class _Mark:
    pass


_SAFE_DTYPES = {"f8", "f4", "i8", "i4", "i2", "i1", "u1", "b1"}


def walk(data):
    """Symbolically evaluate the opcode stream of ``data`` (bytes)."""
    stack, memo = [], {}

    def pop_to_mark():
        items = []
        while True:
            x = stack.pop()
            if isinstance(x, _Mark):
                break
            items.append(x)
        items.reverse()
        return items

    for op, arg, _pos in pickletools.genops(data):
        n = op.name
        if n in ("PROTO", "FRAME"):
            continue
        elif n == "STOP":
            break
        elif n == "GLOBAL":
            stack.append(Global(arg.replace(" ", ".")))
        elif n == "STACK_GLOBAL":
            name = stack.pop()
            mod = stack.pop()
            stack.append(Global("%s.%s" % (mod, name)))
        elif n in ("BINPUT", "LONG_BINPUT", "PUT"):
            memo[arg] = stack[-1]
        elif n == "MEMOIZE":
            memo[len(memo)] = stack[-1]
        elif n in ("BINGET", "LONG_BINGET", "GET"):
            stack.append(memo[arg])
        elif n == "MARK":
            stack.append(_Mark())
        elif n == "EMPTY_TUPLE":
            stack.append(())
        elif n == "EMPTY_DICT":
            stack.append({})
        elif n == "EMPTY_LIST":
            stack.append([])
        elif n in ("BINUNICODE", "SHORT_BINUNICODE", "BINUNICODE8", "UNICODE",
                   "BININT", "BININT1", "BININT2", "LONG1", "LONG4", "INT", "LONG",
                   "BINFLOAT", "FLOAT", "BINBYTES", "SHORT_BINBYTES", "BINBYTES8",
                   "BINSTRING", "SHORT_BINSTRING"):
            stack.append(arg)
        elif n == "BYTEARRAY8":
            stack.append(bytes(arg))
        elif n == "NONE":
            stack.append(None)
        elif n == "NEWTRUE":
            stack.append(True)
        elif n == "NEWFALSE":
            stack.append(False)
        elif n == "TUPLE1":
            a = stack.pop()
            stack.append((a,))
        elif n == "TUPLE2":
            b = stack.pop(); a = stack.pop()
            stack.append((a, b))
        elif n == "TUPLE3":
            c = stack.pop(); b = stack.pop(); a = stack.pop()
            stack.append((a, b, c))
        elif n == "TUPLE":
            stack.append(tuple(pop_to_mark()))
        elif n == "LIST":
            stack.append(list(pop_to_mark()))
        elif n == "DICT":
            it = pop_to_mark()
            stack.append(dict(zip(it[0::2], it[1::2])))
        elif n == "APPEND":
            v = stack.pop()
            stack[-1].append(v)
        elif n == "APPENDS":
            it = pop_to_mark()
            stack[-1].extend(it)
        elif n == "SETITEM":
            v = stack.pop(); k = stack.pop()
            stack[-1][k] = v
        elif n == "SETITEMS":
            it = pop_to_mark()
            d = stack[-1]
            for k, v in zip(it[0::2], it[1::2]):
                d[k] = v
        elif n in ("NEWOBJ", "REDUCE"):
            args = stack.pop()
            callee = stack.pop()
            stack.append(Inert(callee, args))
        elif n == "BUILD":
            state = stack.pop()
            tgt = stack[-1]
            if not isinstance(tgt, Inert):
                raise ValueError("BUILD on non-record")
            tgt.state = state
        else:
            raise ValueError("opcode %s is not handled by the static reader" % n)
    return stack[-1]


def _dtype_of(node):
    if not (isinstance(node, Inert) and isinstance(node.callee, Global)
            and node.callee.name == "numpy.dtype"):
        raise ValueError("not a dtype record")
    descr = node.args[0]
    if descr not in _SAFE_DTYPES:
        raise ValueError("dtype %r not whitelisted" % (descr,))
    order = node.state[1] if node.state else "<"
    if order not in ("<", "|", "="):
        raise ValueError("unexpected byte order %r" % (order,))
    return np.dtype(("<" if order != "|" else "|") + descr)


def _raw_bytes(node):
    """Byte payload of an array: bytes, or the `_codecs.encode(text, 'latin1')` record that protocols
    below 3 use for bytes."""
    if isinstance(node, (bytes, bytearray)):
        return bytes(node)
    if (isinstance(node, Inert) and isinstance(node.callee, Global) and node.callee.name == "_codecs.encode"
            and len(node.args) == 2 and isinstance(node.args[0], str) and node.args[1] == "latin1"):
        return node.args[0].encode("latin1")
    raise ValueError("array payload is not a byte string")


def decode(node):
    """Inert tree -> plain python/numpy data.  Objects become dicts of their state
    under the key ``__state__`` with ``__class__`` holding the class *name*."""
    if isinstance(node, Inert):
        name = node.callee.name if isinstance(node.callee, Global) else None
        if name in ("numpy.core.multiarray._reconstruct", "numpy._core.multiarray._reconstruct"):
            _ver, shape, dt, fortran, raw = node.state
            arr = np.frombuffer(_raw_bytes(raw), dtype=_dtype_of(dt))
            arr = arr.reshape(shape, order="F" if fortran else "C")
            return arr.copy()
        if name in ("numpy.core.numeric._frombuffer", "numpy._core.numeric._frombuffer"):
            raw, dt, shape, order = node.args
            if order not in ("C", "F"):
                raise ValueError("unexpected array order %r" % (order,))
            return np.frombuffer(_raw_bytes(raw), dtype=_dtype_of(dt)).reshape(shape, order=order).copy()
        if name in ("numpy.core.multiarray.scalar", "numpy._core.multiarray.scalar"):
            dt, raw = node.args
            return np.frombuffer(_raw_bytes(raw), dtype=_dtype_of(dt))[0]
        if name == "numpy.dtype":
            return str(_dtype_of(node))
        return {"__class__": name,
                "__state__": decode(node.state) if node.state is not None else None}
    if isinstance(node, dict):
        return {decode(k) if not isinstance(k, str) else k: decode(v) for k, v in node.items()}
    if isinstance(node, list):
        return [decode(v) for v in node]
    if isinstance(node, tuple):
        return tuple(decode(v) for v in node)
    return node


def read(path):
    with open(path, "rb") as f:
        return decode(walk(f.read()))
