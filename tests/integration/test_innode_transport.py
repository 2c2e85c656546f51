"""The Coordinator over the in-node transport (no sockets, no JSON)."""

import asyncio

import pytest
import torch

from nanofed_amd import Coordinator, CoordinatorConfig, FedAvgAggregator, ModelManager, coordinate
from nanofed_amd.communication import InNodeClient, InNodeServer
from nanofed_amd.core import NanoFedError
from nanofed_amd.models import MNISTMLP
from nanofed_amd.runtime import ParameterArena


async def client_loop(cid, server, samples, shift, as_arena):
    async with InNodeClient(server.url, cid) as client:
        done = set()
        while not await client.check_server_status():
            state, rnd = await client.fetch_global_model()
            if rnd in done:
                await asyncio.sleep(0.01)
                continue
            model = MNISTMLP(hidden=8)
            model.load_state_dict(state)
            with torch.no_grad():
                for p in model.parameters():
                    p.add_(shift)
            payload = ParameterArena.from_module(model).views() if as_arena else model
            if await client.submit_update(payload, {"loss": 1.0, "samples_processed": samples}):
                done.add(rnd)


@pytest.mark.parametrize("as_arena", [False, True])
def test_coordinator_round_trip_over_the_in_node_transport(tmp_path, as_arena):
    async def go():
        server = InNodeServer("t")
        await server.start()
        torch.manual_seed(0)
        manager = ModelManager(MNISTMLP(hidden=8))
        start = {k: v.clone() for k, v in manager.model.state_dict().items()}
        coord = Coordinator(manager, FedAvgAggregator(), server,
                            CoordinatorConfig(2, 2, 1.0, 30, tmp_path))
        await asyncio.gather(
            coordinate(coord),
            client_loop("a", server, 300, 1.0, as_arena),
            client_loop("b", server, 100, -1.0, as_arena),
        )
        await server.stop()
        return start, manager
    start, manager = asyncio.run(go())
    # each round moves every parameter by 0.75 * 1 + 0.25 * (-1) = +0.5
    for k, v in manager.model.state_dict().items():
        torch.testing.assert_close(v, start[k] + 1.0, rtol=1e-5, atol=1e-5)


def test_protocol_rules_match_the_http_server(tmp_path):
    async def go():
        server = InNodeServer("rules")
        await server.start()
        async with InNodeClient(server, "c") as client:
            with pytest.raises(NanoFedError, match="Server not initialized with coordinator"):
                await client.fetch_global_model()
            Coordinator(ModelManager(MNISTMLP(hidden=4)), FedAvgAggregator(), server,
                        CoordinatorConfig(1, 1, 1.0, 5, tmp_path))
            state, rnd = await client.fetch_global_model()
            assert rnd == 0 and set(state) == {"fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias"}
            server.advance_round(3)
            with pytest.raises(NanoFedError, match="Invalid round number"):
                await client.submit_update(MNISTMLP(hidden=4), {"loss": 0.0})
            await server.stop_training()
            assert await client.check_server_status() is True
            with pytest.raises(NanoFedError, match="Training is complete"):
                await client.fetch_global_model()
            assert await client.submit_update(MNISTMLP(hidden=4), {"loss": 0.0}) is False
        with pytest.raises(NanoFedError):
            async with InNodeClient("innode://nowhere", "x") as c2:
                await c2.fetch_global_model()
    asyncio.run(go())
