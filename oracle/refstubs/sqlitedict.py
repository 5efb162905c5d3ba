"""Minimal `sqlitedict.SqliteDict` stand-in over sqlite3 (pyprob/util.py:347-355 opens shelves with it).
Only what shelve.Shelf needs. Test infrastructure only."""
import sqlite3
from pickle import dumps, loads, HIGHEST_PROTOCOL


def _enc(obj):
    return sqlite3.Binary(dumps(obj, protocol=HIGHEST_PROTOCOL))


def _dec(obj):
    return loads(bytes(obj))


class SqliteDict:
    def __init__(self, filename=None, tablename='unnamed', flag='c', autocommit=False, journal_mode='DELETE',
                 encode=_enc, decode=_dec, timeout=5, outer_stack=True):
        self.filename = filename
        self.encode, self.decode = encode, decode
        self.conn = sqlite3.connect(filename)
        self.conn.execute('CREATE TABLE IF NOT EXISTS "unnamed" (key TEXT PRIMARY KEY, value BLOB)')
        self.conn.commit()

    def __len__(self):
        return self.conn.execute('SELECT COUNT(*) FROM "unnamed"').fetchone()[0]

    def __contains__(self, key):
        return self.conn.execute('SELECT 1 FROM "unnamed" WHERE key = ?', (key,)).fetchone() is not None

    def __getitem__(self, key):
        row = self.conn.execute('SELECT value FROM "unnamed" WHERE key = ?', (key,)).fetchone()
        if row is None:
            raise KeyError(key)
        return self.decode(row[0])

    def __setitem__(self, key, value):
        self.conn.execute('REPLACE INTO "unnamed" (key, value) VALUES (?,?)', (key, self.encode(value)))

    def __delitem__(self, key):
        self.conn.execute('DELETE FROM "unnamed" WHERE key = ?', (key,))

    def keys(self):
        return [r[0] for r in self.conn.execute('SELECT key FROM "unnamed"')]

    def __iter__(self):
        return iter(self.keys())

    def sync(self):
        self.conn.commit()

    commit = sync

    def close(self, *a, **k):
        self.conn.commit()
        self.conn.close()
