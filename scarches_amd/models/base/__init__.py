from ._base import BaseMixin, SurgeryMixin, CVAELatentsMixin

__all__ = ["BaseMixin", "SurgeryMixin", "CVAELatentsMixin"]
