"""The sample-metadata database of an index: `<basename>.meta.mor` (sqlite).

Mirrors commanderson/morna morna.py:494-520 (written by MornaIndex.save when a
metafile was given) and morna.py:666-676 / 718-728 / 775-785 (the join the three
search methods do when `meta_db=True`).  Host-only string work; nothing here
touches the GPU.

Metafile format: one sample per line, whitespace separated; first column the
sample id, the rest of the line (trailing newline included, as the reference
stores it) the keywords.
"""
import sqlite3


def write_meta_db(metafile, basename):
    """(Re)create table metadata(sample_id real, keywords text) from `metafile` (morna.py:494-520)."""
    conn = sqlite3.connect(basename + '.meta.mor')
    cursor = conn.cursor()
    cursor.execute("SELECT name FROM sqlite_master WHERE type='table' AND name='metadata'")
    if cursor.fetchone():
        cursor.execute("DROP TABLE metadata")          # overwriting an old index
    cursor.execute("CREATE TABLE metadata (sample_id real, keywords text)")
    with open(metafile) as metafile_handle:
        for line in metafile_handle:
            fields = line.split(None, 1)
            # the reference formats both fields into the statement as quoted text; bound
            # parameters store the same values (and survive a quote in the keywords)
            cursor.execute("INSERT INTO metadata VALUES (?,?)", (fields[0], fields[1]))
    conn.commit()
    conn.close()


def lookup_meta(basename, sample_ids):
    """One `cursor.fetchone()` per sample id: a 1-tuple `(keywords,)` or None (morna.py:667-675)."""
    meta_results = ['' for _ in sample_ids]
    conn = sqlite3.connect(basename + ".meta.mor")
    cursor = conn.cursor()
    for i, sample_id in enumerate(sample_ids):
        cursor.execute('SELECT keywords FROM metadata WHERE sample_id=?', (str(sample_id),))
        meta_results[i] = cursor.fetchone()
    conn.close()
    return meta_results
