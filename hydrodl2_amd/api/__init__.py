from .methods import available_models, available_modules, load_model, load_module

__all__ = ["available_models", "available_modules", "load_model", "load_module"]
