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


def junctions_from_bed_stream(bed_stream):
    """BED12 lines of a junctions file (utils.py:206-252)."""
    for line in bed_stream:
        tokens = line.rstrip().split('\t')
        if len(tokens) < 12:
            continue
        chrom = tokens[0]
        chrom_start = int(tokens[1])
        coverage = int(tokens[4])
        block_sizes = tokens[10].split(',')
        block_starts = tokens[11].split(',')
        # Handle trailing commas
        try:
            int(block_sizes[-1])
        except ValueError:
            block_sizes = block_sizes[:-1]
        try:
            int(block_starts[-1])
        except ValueError:
            block_starts = block_starts[:-1]
        block_count = len(block_sizes)
        if block_count < 2:
            continue
        assert block_count == len(block_starts)
        junctions = [chrom_start + int(block_starts[0]) + int(block_sizes[0])]
        for i in range(1, block_count - 1):
            junction_start = chrom_start + int(block_starts[i])
            junctions.append(junction_start)
            junctions.append(junction_start + int(block_sizes[i]))
        junctions.append(chrom_start + int(block_starts[-1]))
        for i in range(len(junctions) // 2):
            yield (chrom, junctions[2 * i] + 1, junctions[2 * i + 1], coverage)


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
