"""The switches of the host side, in ONE registry.

Rounds 1-5 grew 19 `os.environ.get("VCVITS_...")` reads across the Python modules and 39 `getenv` calls in the kernel files.
Now:
  * the kernel files read ONE struct (csrc/tuning.h: `VcvTuning`, defaults baked in), initialised from
    VCVITS_TUNING="key=value,..." and reachable at run time through `kernel_get` / `kernel_set` below (C ABI:
    vcv_tuning_get / vcv_tuning_set);
  * every environment variable the Python side honours is declared through `flag` / `integer` / `number` / `text` HERE --
    name, default, one line of meaning -- and nowhere else is `os.environ` consulted for a VCVITS_* name.  The modules keep
    their one-element-list switches (shared by reference across the ops package), initialised from these declarations.

`python -m vcvits_amd.tuning` prints both tables (DESIGN section 10 points here)."""
import os

REGISTRY = {}  # name -> (kind, default, doc, value)


def _declare(name, kind, default, doc, value):
    REGISTRY[name] = (kind, default, doc, value)
    return value


def flag(name, default, doc):
    """A 0 / 1 environment switch."""
    raw = os.environ.get(name)
    return _declare(name, "flag", default, doc, default if raw is None else raw == "1")


def integer(name, default, doc):
    raw = os.environ.get(name)
    return _declare(name, "int", default, doc, default if raw is None else int(raw))


def number(name, default, doc):
    raw = os.environ.get(name)
    return _declare(name, "float", default, doc, default if raw is None else float(raw))


def text(name, default, doc):
    return _declare(name, "text", default, doc, os.environ.get(name, default))


def live_flag(name):
    """Re-read a declared flag from the environment NOW (the few switches tests flip with monkeypatch.setenv after import)."""
    kind, default, doc, _ = REGISTRY[name]
    raw = os.environ.get(name)
    return default if raw is None else (raw == "1" if kind == "flag" else int(raw))


# ---- the kernel-side table (csrc/tuning.h) ---------------------------------------------------------------------------------
KERNEL_KEYS = ("xcd_remap", "pk_ws", "pk_ws_bf16", "pk_x4", "pk_vec", "x3_variant", "x3_v6", "x3_js2", "x3_old_ks", "x3_all", "x3_terms",
               "wgrad_dma", "wgrad_tile", "wgrad_verbose", "wgrad_bf16_ws", "wgrad_finish_vec", "bias_rows", "c1_chunk", "m1_lds",
               "c1_wgrad_pairs", "thin_wgrad_wgs", "act_grad_vec", "ln_regs", "stft_wave", "attn_rows", "zero_memset", "pack_tile",
               "pack_tile_bf16", "pair_dbg", "pair_grid", "pair_stream", "deterministic")


def kernel_get(key):
    import ctypes
    from ._lib import check, lib
    v = ctypes.c_int(0)
    check(lib().vcv_tuning_get(key.encode(), ctypes.byref(v)), "vcv_tuning_get(%s)" % key)
    return v.value


def kernel_set(key, value):
    from ._lib import check, lib
    check(lib().vcv_tuning_set(key.encode(), int(value)), "vcv_tuning_set(%s)" % key)


def table():
    """Both tables as text lines."""
    import importlib
    for mod in ("vcvits_amd._lib", "vcvits_amd.ops", "vcvits_amd.mel_processing", "vcvits_amd.model.flow", "vcvits_amd.model.modules",
                "vcvits_amd.model.discriminators._pair", "vcvits_amd.light.optim", "vcvits_amd.light.graphed"):
        importlib.import_module(mod)  # (the declarations run at import)
    out = ["environment variables of the host side (name, default, meaning):"]
    for name in sorted(REGISTRY):
        kind, default, doc, value = REGISTRY[name]
        d = ("1" if default else "0") if kind == "flag" else str(default)
        out.append("  %-28s %-10s %s" % (name, d, doc))
    out.append("")
    out.append("kernel tuning table (VCVITS_TUNING=\"key=value,...\"; csrc/tuning.h documents each key):")
    try:
        out.append("  " + ", ".join("%s=%d" % (k, kernel_get(k)) for k in KERNEL_KEYS))
    except Exception as e:  # noqa: BLE001  (no library built)
        out.append("  (library not loaded: %s)" % e)
    return out


if __name__ == "__main__":
    from vcvits_amd import tuning as _t  # (the package's instance of this module holds the registry, not __main__'s)
    print("\n".join(_t.table()))
