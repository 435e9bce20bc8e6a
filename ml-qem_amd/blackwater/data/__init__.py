"""Data layer of the path: circuit IR, encoders, entry record, dataset, graph containers."""
