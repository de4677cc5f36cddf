def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X")
    config.addinivalue_line("markers", "experiment: a measured-and-rejected kernel of tools/experiments (not part of the product test run)")
