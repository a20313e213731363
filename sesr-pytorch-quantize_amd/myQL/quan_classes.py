"""Config holders used when splicing callables into a model graph.

Same vocabulary as the reference (myQL/quan_classes.py:9-38): a FunctionPackage pairs a callable
with its keyword arguments, a NodeInsertMappingElement binds it to a module type, a
NodeInsertMapping collects such bindings, a NodeInsertConfig is the per-node lookup result."""
from dataclasses import dataclass, field
from typing import Any, Callable, Dict, Optional, Tuple, Type


@dataclass
class FunctionPackage:
    function: Callable
    parameter_dict: Optional[Dict[str, Any]] = None

    def __init__(self, the_function: Callable, parameter_dict: Optional[Dict[str, Any]] = None):
        self.function = the_function
        self.parameter_dict = parameter_dict


@dataclass
class NodeInsertConfig:
    should_insert: bool
    function_package: Optional[FunctionPackage] = None


class NodeInsertMappingElement:
    def __init__(self, insert_type: Type, the_function: FunctionPackage):
        self.insert_mapping_config: Tuple[Type, FunctionPackage] = (insert_type, the_function)

    def get_config(self) -> Tuple[Type, FunctionPackage]:
        return self.insert_mapping_config


class NodeInsertMapping:
    def __init__(self):
        self.insert_mapping: Dict[Type, FunctionPackage] = {}

    def add_config(self, insert_mapping_config: NodeInsertMappingElement) -> None:
        module_type, package = insert_mapping_config.get_config()
        self.insert_mapping[module_type] = package

    def get_mapping(self) -> Dict[Type, FunctionPackage]:
        return self.insert_mapping
