"""Text readers for the interaction formats that feed this path — the `Reader` of the reference
(cornac/data/reader.py:101-356) for its 'UI', 'UIR' and 'UIRT' line formats: same constructor arguments, same
filters applied in the same order, same returned list of tuples, so `Dataset.from_uir(Reader().read(path))` reads the
same files the reference's examples read.  The basket / sequence / review formats belong to model families outside
this path and are rejected with the reference's error message shape."""
import itertools
from collections import Counter

FORMATS = ("UI", "UIR", "UIRT")


def _parse_ui(tokens, line_idx, id_inline):
    # one implicit-feedback tuple per listed item; the user is the first token, or the 1-based line number
    user, items = (str(line_idx + 1), tokens) if id_inline else (tokens[0], tokens[1:])
    return [(user, item, 1.0) for item in items]


def _parse_uir(tokens, line_idx, id_inline):
    return [(tokens[0], tokens[1], float(tokens[2]))]


def _parse_uirt(tokens, line_idx, id_inline):
    return [(tokens[0], tokens[1], float(tokens[2]), int(tokens[3]))]


_PARSERS = {"UI": _parse_ui, "UIR": _parse_uir, "UIRT": _parse_uirt}


class Reader:
    def __init__(self, user_set=None, item_set=None, min_user_freq=1, min_item_freq=1, num_top_freq_user=0,
                 num_top_freq_item=0, bin_threshold=None, encoding="utf-8", errors=None):
        self.user_set = None if user_set is None else set(user_set)
        self.item_set = None if item_set is None else set(item_set)
        self.min_uf, self.min_if = min_user_freq, min_item_freq
        self.num_top_freq_user, self.num_top_freq_item = num_top_freq_user, num_top_freq_item
        self.bin_threshold = bin_threshold
        self.encoding, self.errors = encoding, errors

    @staticmethod
    def _keep(tuples, col, allowed):
        return [t for t in tuples if t[col] in allowed]

    def _filter(self, tuples, fmt):
        """reader.py:207-245: binarise, most frequent users, most frequent items, allowed users, allowed items,
        minimum user frequency, minimum item frequency — each stage counting on the output of the previous one"""
        U, I, R = 0, 1, fmt.find("R")
        if self.bin_threshold is not None and R >= 0:
            tuples = [t[:R] + (1.0,) + t[R + 1:] for t in tuples if t[R] >= self.bin_threshold]
        for col, top in ((U, self.num_top_freq_user), (I, self.num_top_freq_item)):
            if top > 0:
                frequent = {key for key, _ in Counter(t[col] for t in tuples).most_common(top)}
                tuples = self._keep(tuples, col, frequent)
        for col, allowed in ((U, self.user_set), (I, self.item_set)):
            if allowed is not None:
                tuples = self._keep(tuples, col, allowed)
        for col, least in ((U, self.min_uf), (I, self.min_if)):
            if least > 1:
                freq = Counter(t[col] for t in tuples)
                tuples = [t for t in tuples if freq[t[col]] >= least]
        return tuples

    def read(self, fpath, fmt="UIR", sep="\t", skip_lines=0, id_inline=False, parser=None, **kwargs):
        if parser is None:
            if fmt not in _PARSERS:
                raise ValueError("Invalid line format: {}\nSupported formats: {}".format(fmt, FORMATS))
            line_parser = _PARSERS[fmt]

            def parse(tokens, idx):
                return line_parser(tokens, idx, id_inline)
        else:   # user-supplied parser: the reference's calling convention (reader.py:330-334)
            def parse(tokens, idx):
                return parser(tokens, line_idx=idx, id_inline=id_inline, **kwargs)
        tuples = []
        with open(fpath, encoding=self.encoding, errors=self.errors) as f:
            for idx, line in enumerate(itertools.islice(f, skip_lines, None)):
                tuples.extend(parse(line.strip().split(sep), idx))
        return self._filter(tuples, fmt)
