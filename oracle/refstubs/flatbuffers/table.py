class Table:
    def __init__(self, buf=None, pos=0):
        self.Bytes = buf
        self.Pos = pos
