"""Query-side junction stream parsers: the counterparts of the reference's
utils.py:194-289 (junctions_from_raw_stream / _bed_stream / _sam_stream).

Each yields (chrom, start, end, coverage) with 1-based inclusive coordinates,
exactly as the reference does.  The SAM parser restates the junction part of
utils.py:37-160 (indels_junctions_exons_mismatches with the dummy MD string of
utils.py:162-192): only reference-consuming CIGAR operations move the position,
an N operation of length n at position p is the junction (p, p + n - 1).
"""
import re
import sys


def junctions_from_raw_stream(raw_stream):
    """chrom <tab> start <tab> end <tab> coverage (utils.py:194-204)."""
    for line in raw_stream:
        tokens = line.strip().split('\t')
        yield (tokens[0], int(tokens[1]), int(tokens[2]), int(tokens[3]))


def _block_field(field):
    """Items of a BED12 blockSizes / blockStarts field.  UCSC writes the lists with a closing comma, so a
    last item that is not a number is not an item; the others are converted where they are used."""
    items = field.split(',')
    try:
        int(items[-1])
    except ValueError:
        items.pop()
    return items


def junctions_from_bed_stream(bed_stream):
    """Introns of BED12 alignment lines (the format utils.py:206-252 reads).

    A BED12 line places n blocks (exons) at chromStart + blockStarts[i], blockSizes[i] long, 0-based half
    open; the n - 1 gaps between consecutive blocks are the junctions, reported 1-based inclusive with the
    line's score column as coverage.  Lines with fewer than 12 columns or fewer than 2 blocks carry none.
    """
    for line in bed_stream:
        columns = line.rstrip().split('\t')
        if len(columns) < 12:
            continue
        origin, coverage = int(columns[1]), int(columns[4])
        sizes, offsets = _block_field(columns[10]), _block_field(columns[11])
        n_blocks = len(sizes)
        if n_blocks < 2:
            continue
        assert n_blocks == len(offsets)
        introns = []
        for left in range(n_blocks - 1):
            last_exon_base = origin + int(offsets[left]) + int(sizes[left])     # 0-based end = 1-based last base
            next_exon_base = origin + int(offsets[left + 1])                     # 0-based start of the next block
            introns.append((last_exon_base + 1, next_exon_base))
        for first, last in introns:
            yield (columns[0], first, last, coverage)


def _cigar_junctions(cigar, pos):
    """(start, end) 1-based inclusive of every N operation (utils.py:103-109, 283-284)."""
    parts = re.split(r'([MINDS])', cigar)[:-1]
    if len(parts) % 2 or any(not p.isdigit() for p in parts[0::2]):
        raise RuntimeError('Accepted CIGAR characters are only in [MINDS].')
    out = []
    for size, op in zip(parts[0::2], parts[1::2]):
        n = int(size)
        if op == 'N':
            out.append((pos, pos + n - 1))
            pos += n
        elif op in 'MD':
            pos += n
        # I and S consume the read only
    return out


def junctions_from_sam_stream(sam_stream):
    """Spliced primary alignments of a SAM stream, coverage 1 each (utils.py:254-289)."""
    for line in sam_stream:
        if line[0] == '@':
            continue
        try:
            tokens = line.strip().split('\t')
            flag = int(tokens[1])
            if flag & 4:
                continue
            rname = tokens[2]
            cigar = tokens[5]
            pos = int(tokens[3])
            tokens[9]                      # the reference reads SEQ; a short line is an IndexError
            if 'N' not in cigar or flag & 256:
                continue
            for start, end in _cigar_junctions(cigar, pos):
                yield (rname, start, end, 1)
        except IndexError:
            sys.stderr.write('Error found on line: ' + line + '\n')
            raise
