"""torch.fx graph splicers with the reference's call surface (myQL/graph_modify.py:33,68,123):

    insert_before(model_input, insert_mapping, has_func_id=None)
    insert_bias_bypass(model_input, insert_mapping)
    insert_after(model_input, insert_mapping)

Each returns a GraphModule in which a call_function node was placed before / after every module
whose type is in the mapping; func_id is the ordinal of the matched module; insert_bias_bypass also
zeroes the module's bias and hands the float bias to the inserted callable as a list.
The returned module is a SesrqGraphModule: its forward lowers the spliced graph to the fused
device op on first use (sesrq/lowering.py)."""
import torch
from torch.fx import symbolic_trace

from myQL.quan_classes import NodeInsertMapping, NodeInsertConfig
from sesrq.lowering import SesrqGraphModule


def get_insert_config(node, modules, node_insert_mapping: NodeInsertMapping) -> NodeInsertConfig:
    if node.op != "call_module":
        return NodeInsertConfig(should_insert=False)
    package = node_insert_mapping.get_mapping().get(type(modules[node.target]))
    return NodeInsertConfig(should_insert=package is not None, function_package=package)


def _traced(model_input):
    return model_input if isinstance(model_input, torch.fx.GraphModule) else symbolic_trace(model_input)


def _splice(model_input, insert_mapping, before, with_id, bypass_bias):
    gm = _traced(model_input)
    modules = dict(gm.named_modules())
    graph = gm.graph
    state = model_input.state_dict() if bypass_bias else None
    function_id = 0
    for node in list(graph.nodes):
        cfg = get_insert_config(node, modules, insert_mapping)
        if not cfg.should_insert:
            continue
        kwargs = dict(cfg.function_package.parameter_dict or {})
        if with_id:
            kwargs["func_id"] = function_id
        if bypass_bias:
            bias = state[node.target + ".bias"]
            kwargs["bias"] = bias.tolist()
            state[node.target + ".bias"] = torch.zeros_like(bias)
        function_id += 1
        if before:
            with graph.inserting_before(node):
                new = graph.call_function(cfg.function_package.function, args=(node.args[0],), kwargs=kwargs)
            node.args = (new,) + tuple(node.args[1:])
        else:
            with graph.inserting_after(node):
                new = graph.call_function(cfg.function_package.function, kwargs=kwargs)
            node.replace_all_uses_with(new)
            new.args = (node,)
    if bypass_bias:
        model_input.load_state_dict(state)
    graph.lint()
    return SesrqGraphModule(model_input, graph)


def insert_before(model_input, insert_mapping: NodeInsertMapping, has_func_id=None) -> torch.fx.GraphModule:
    return _splice(model_input, insert_mapping, before=True, with_id=has_func_id is not None, bypass_bias=False)


def insert_after(model_input, insert_mapping: NodeInsertMapping) -> torch.fx.GraphModule:
    return _splice(model_input, insert_mapping, before=False, with_id=True, bypass_bias=False)


def insert_bias_bypass(model_input, insert_mapping: NodeInsertMapping) -> torch.fx.GraphModule:
    return _splice(model_input, insert_mapping, before=False, with_id=True, bypass_bias=True)
