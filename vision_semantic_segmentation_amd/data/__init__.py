"""Host-side tables (confusion matrix) consumed by the grid-update kernel."""
