"""A thin HTTP front for an index served by rad_amd.

Answers the routes a `RemoteHNSWService` client of the reference talks to, with the same JSON shapes
(reference: rad/hnsw_server.py — `/neighbors/{node_id}/{level}` :451-515 returns
{"node_id", "level", "neighbors": [id, smiles, ...], "neighbor_count", "request_id"};
`/top-level-nodes` :517-547 returns {"top_nodes", "node_count", "cached", "request_id"};
`/health` :549-582; `/info` :584-619; `/ping` :447), plus one batched route for SURVEY.md §8f N2.

It is transport only: every request is one call into `rad_amd.hnsw_service.LocalHNSWService`, whose
adjacency reads go straight into librad_hip's host mirror of the graph (no process hop, no GPU needed:
the reference's production server also loads the index without vectors, scripts/start_hnsw_server.py:69).
Out of scope here as in the rest of rad_amd: the landing page / static files, CORS, the metrics class.
"""
from __future__ import annotations

import time
import uuid
from typing import List, Optional

from .hnsw_service import LocalHNSWService


def create_app(hnsw, database_path: Optional[str] = None, api_key: Optional[str] = None):
    """FastAPI app serving `hnsw` (a rad_amd.Index or anything with the duck-typed usearch surface)."""
    from fastapi import Body, Depends, FastAPI, Header, HTTPException

    service = hnsw if isinstance(hnsw, LocalHNSWService) else LocalHNSWService(hnsw, database_path=database_path)
    index = service.hnsw
    app = FastAPI(title="RAD HNSW Service (rad_amd)", version="1.0.0")
    started = time.time()
    top_cache: List = service.get_top_level_nodes()       # computed once, as the reference's start-up cache

    def auth(authorization: Optional[str] = Header(default=None)):
        if api_key is None:
            return None
        if authorization != f"Bearer {api_key}":
            raise HTTPException(status_code=401, detail="Invalid or missing API key")
        return api_key

    def rid() -> str:
        return str(uuid.uuid4())

    def check(node_id: int, level: int) -> None:
        if node_id < 0:
            raise HTTPException(status_code=400, detail="node_id must be non-negative")
        if level < 0:
            raise HTTPException(status_code=400, detail="level must be non-negative")
        size = len(index)
        if node_id >= size:
            raise HTTPException(status_code=400, detail=f"node_id {node_id} is out of range (max: {size - 1})")
        if level > index.max_level:
            raise HTTPException(status_code=400, detail=f"level {level} is out of range (max: {index.max_level})")

    @app.get("/ping")
    def ping():
        return {"pong": True}

    @app.get("/neighbors/{node_id}/{level}")
    def neighbors(node_id: int, level: int, _k=Depends(auth)):
        check(node_id, level)
        try:
            nb = service.get_neighbors(node_id, level)
        except RuntimeError:
            raise HTTPException(status_code=400, detail=f"Invalid node_id/level combination: node {node_id} may not exist at level {level}")
        return {"node_id": node_id, "level": level, "neighbors": nb, "neighbor_count": len(nb) // 2, "request_id": rid()}

    @app.post("/neighbors-many")
    def neighbors_many(body: dict = Body(...), _k=Depends(auth)):
        """{"pairs": [[node_id, level], ...]} -> one row per pair; one SMILES query for all of them"""
        try:
            pairs = [(int(p[0]), int(p[1])) for p in body["pairs"]]
        except Exception:
            raise HTTPException(status_code=400, detail='body must be {"pairs": [[node_id, level], ...]}')
        for n, lv in pairs:
            check(n, lv)
        try:
            rows = service.get_neighbors_many(pairs)
        except RuntimeError as e:
            raise HTTPException(status_code=400, detail=str(e))
        return {"results": [{"node_id": n, "level": lv, "neighbors": r, "neighbor_count": len(r) // 2}
                            for (n, lv), r in zip(pairs, rows)], "request_id": rid()}

    @app.get("/top-level-nodes")
    def top_level_nodes(_k=Depends(auth)):
        return {"top_nodes": top_cache, "node_count": len(top_cache) // 2, "cached": True, "request_id": rid()}

    @app.get("/health")
    def health():
        return {"status": "healthy" if service.is_healthy() else "unhealthy", "timestamp": time.time(), "hnsw_size": int(len(index)),
                "hnsw_max_level": int(index.max_level), "uptime_seconds": time.time() - started, "request_id": rid()}

    @app.get("/info")
    def info(_k=Depends(auth)):
        return {"service_type": "RemoteHNSWService", "version": "1.0.0", "hnsw_info": service.get_hnsw_info(),
                "performance_metrics": {"total_requests": service.request_count, "total_errors": service.error_count,
                                        "uptime_seconds": time.time() - started},
                "authentication_enabled": api_key is not None, "cors_enabled": False, "debug_mode": False, "request_id": rid()}

    return app


def run_hnsw_server(hnsw, host: str = "0.0.0.0", port: int = 8000, **kwargs) -> None:
    """reference: rad/hnsw_server.py:652-675"""
    import uvicorn
    uvicorn.run(create_app(hnsw, **kwargs), host=host, port=port)


__all__ = ["create_app", "run_hnsw_server"]
