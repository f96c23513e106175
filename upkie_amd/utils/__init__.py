"""Host-side helpers with the reference's names (upkie/utils/*)."""
