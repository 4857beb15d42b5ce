"""Field naming of the key-value head (reference: inference/postprocess.py:1-15)."""

CLASS_NAMES = ["NUL"] + [f"{kind}_{field}" for field in (
    "bank_name", "bank_branch_name", "account_number", "account_type", "account_name", "account_name_kana", "branch",
    "financial_institution") for kind in ("k", "v")]


def post_process_kv(values):
    """values[idx] = (text, boxes, intersect, union) per class -> {field name: text} for the value classes.
    Value classes are the odd indices >= 3; class idx is named after CLASS_NAMES[idx - 1] without its 'k_'/'v_'
    prefix, or str(idx - 1) past the table (postprocess.py:8-15)."""
    out = {}
    for idx in range(3, len(values), 2):
        name = CLASS_NAMES[idx - 1][2:] if idx - 1 < len(CLASS_NAMES) else str(idx - 1)
        out[name] = values[idx][0]
    return out
