from oracle_amd.scene_graph_prediction.main import config_loader, main  # noqa: F401

if __name__ == '__main__':
    main()
