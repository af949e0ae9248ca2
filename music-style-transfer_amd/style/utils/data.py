"""CSV logging of the training run (style/utils/data.py:27-46)."""
import csv
import os

from style.utils.misc import assert_dir


def save_to_csv(path, data=(), fieldnames=None, when_exists='append', **row):
    """Append `row` (and/or the dicts in `data`) to `path`, writing the header when the file is new."""
    fieldnames = fieldnames or list(row.keys())
    if when_exists == 'append':
        mode, header = 'at', not os.path.isfile(path)
    elif when_exists == 'overwrite':
        mode, header = 'wt', True
    else:
        raise Exception(f"Unknown option: {when_exists}")
    assert_dir(path)
    with open(path, mode, encoding='utf-8') as f:
        writer = csv.DictWriter(f, fieldnames)
        if header:
            writer.writeheader()
        if row:
            writer.writerow(row)
        for d in data:
            writer.writerow(d)
