"""Host-side mirror of the reference's ``mdt.models`` operator API for the action-denoising hot path.

A Hydra tree switches to this implementation by re-targeting two strings
(``mdt.models.…`` -> ``mdt_policy_amd.models.…``); see INTEGRATION.md.
"""
