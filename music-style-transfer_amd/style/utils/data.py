"""Tabular host helpers the reference's callers import from `style.utils.data` (train-model.py:29 takes `save_to_csv` and
`assert_dir`, style/data.py:16 takes `list2df`).  Own implementations on the csv module; pandas is imported only when a
caller asks `list2df` for a DataFrame.  The sync-free training loop writes its loss log through `style.train.LossLog`
(fixed columns, many rows per call); `save_to_csv` is the reference's one-row-per-call form of the same file."""
import csv
import os

from style.utils.misc import assert_dir  # noqa: F401  (re-exported: train-model.py:29 imports it from here)

__all__ = ['save_to_csv', 'assert_dir', 'list2df']


def save_to_csv(path, data=(), fieldnames=None, when_exists='append', **row):
    """Append (`when_exists='append'`) or rewrite (`'overwrite'`) a CSV file: the keyword row first, then every dict of
    `data`.  The header is written whenever the call starts the file (it did not exist, or is being overwritten);
    columns are `fieldnames`, by default the keyword row's keys in call order (train-model.py:149)."""
    if when_exists not in ('append', 'overwrite'):
        raise ValueError(f'unknown when_exists option: {when_exists!r}')
    rows = ([row] if row else []) + list(data)
    if fieldnames is None:
        fieldnames = list(row) if row else (list(rows[0]) if rows else [])
    starts_file = when_exists == 'overwrite' or not os.path.isfile(path)
    assert_dir(path)
    with open(path, 'w' if when_exists == 'overwrite' else 'a', newline='', encoding='utf-8') as f:
        out = csv.DictWriter(f, fieldnames)
        if starts_file:
            out.writeheader()
        out.writerows(rows)


def _flat_path(d, prefix=()):
    flat = {}
    for k, v in d.items():
        if isinstance(v, dict):
            flat.update(_flat_path(v, prefix + (k,)))
        else:
            flat[os.path.join(*prefix, k) if prefix else k] = v
    return flat


def list2df(lst, flatten=False, recursive=(), columns=(), include_all_columns=False):
    """Records (a list of dicts) as a pandas DataFrame, optionally with nested dicts flattened to path-like column names,
    nested record lists (`recursive` columns) converted too, and the columns selected / ordered."""
    import pandas as pd
    records = [_flat_path(d) for d in lst] if flatten else list(lst)
    df = pd.DataFrame.from_records(records)
    for col in recursive:
        df[col] = [list2df(v, flatten=flatten) for v in df[col]]
    if columns:
        wanted = list(columns)
        if include_all_columns:
            wanted += [c for c in df.columns if c not in wanted]
        df = df[wanted]
    return df
