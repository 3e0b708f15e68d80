"""CSV logger with the reference's interface (reference uresnet/utils.py:13-44): header from
the first write(), every value formatted '{:f}'."""


class CSVData:
    def __init__(self, fout):
        self._fout = fout
        self._str = None
        self._dict = {}

    def record(self, keys, vals):
        for i, key in enumerate(keys):
            self._dict[key] = vals[i]

    def write(self):
        if self._str is None:
            self._fout = open(self._fout, 'w')
            self._fout.write(','.join(self._dict.keys()) + '\n')
            self._str = ','.join(['{:f}'] * len(self._dict)) + '\n'
        self._fout.write(self._str.format(*(self._dict.values())))

    def flush(self):
        if self._str is not None:
            self._fout.flush()

    def close(self):
        if self._str is not None:
            self._fout.close()


def round_decimals(val, digits):
    factor = float(10 ** digits)
    return int(val * factor + 0.5) / factor
