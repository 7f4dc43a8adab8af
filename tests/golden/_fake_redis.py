"""In-memory stand-in for the `redis` Python client, used ONLY by make_golden.py to drive the
reference's own control flow (rad/coordination_service.py, rad/distributed_worker.py,
rad/traverser.py) in a container that has neither redis-py nor redis-server.

It restates documented Redis semantics for exactly the commands the reference issues:
  ZADD / ZCARD / ZRANGE 0 0 WITHSCORES + ZREM  (sorted set: ascending score, ties by bytewise
                                                member order; ZADD on an existing member
                                                overwrites its score)
  SISMEMBER / SADD, HEXISTS / HSET / HGET / HDEL / HKEYS, RPUSH / LRANGE / LLEN, PING
and recognises the reference's three Lua scripts by the commands they contain
(rad/priority_queue.py:22-27, rad/visited.py:17-24, rad/scored.py:37-47).
Replies are bytes, as with decode_responses=False (rad/traverser.py:92-96).
"""


def _b(x):
    if isinstance(x, bytes):
        return x
    if isinstance(x, float):
        return repr(x).encode()
    return str(x).encode()


class _Script:
    def __init__(self, client, text):
        self.c = client
        if "ZRANGE" in text and "ZREM" in text:
            self.kind = "pop"
        elif "SISMEMBER" in text and "SADD" in text:
            self.kind = "visited"
        elif "HEXISTS" in text and "RPUSH" in text:
            self.kind = "scored"
        else:
            raise NotImplementedError("unknown Lua script")

    def __call__(self, keys=(), args=()):
        c = self.c
        if self.kind == "pop":
            z = c.zsets.get(keys[0], {})
            if not z:
                return None
            member = min(z, key=lambda m: (z[m], m))
            score = z.pop(member)
            # Redis replies with the shortest round-tripping decimal of the double
            return [member, repr(float(score)).encode()]
        if self.kind == "visited":
            s = c.sets.setdefault(keys[0], set())
            m = _b(args[0])
            if m in s:
                return 1
            s.add(m)
            return 0
        if self.kind == "scored":
            node_id, score, smiles = (_b(a) for a in args)
            h = c.hashes.setdefault(keys[1], {})
            if node_id not in h:
                h[node_id] = score
                c.hashes.setdefault(keys[2], {})[node_id] = smiles
                c.lists.setdefault(keys[0], []).append(node_id)
            return None


class StrictRedis:
    def __init__(self, *a, **kw):
        self.zsets, self.sets, self.hashes, self.lists = {}, {}, {}, {}

    def ping(self):
        return True

    def register_script(self, text):
        return _Script(self, text)

    def zadd(self, name, mapping):
        z = self.zsets.setdefault(name, {})
        for m, s in mapping.items():
            z[_b(m)] = float(s)

    def zcard(self, name):
        return len(self.zsets.get(name, {}))

    def hset(self, name, key, value):
        self.hashes.setdefault(name, {})[_b(key)] = _b(value)

    def hget(self, name, key):
        return self.hashes.get(name, {}).get(_b(key))

    def hdel(self, name, key):
        self.hashes.get(name, {}).pop(_b(key), None)

    def hkeys(self, name):
        return list(self.hashes.get(name, {}).keys())

    def lrange(self, name, start, end):
        lst = self.lists.get(name, [])
        return lst[start:] if end == -1 else lst[start:end + 1]

    def llen(self, name):
        return len(self.lists.get(name, []))

    def shutdown(self, *a, **kw):
        pass


Redis = StrictRedis
